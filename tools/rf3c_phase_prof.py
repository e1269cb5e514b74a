"""Cycles per phase of ly_rf3c_bwd pass C (block 0, thread 0) for one launch inside a bf16 lead-yolo-s bs=64 training step.  Needs a library
built with -DRC_PHASE_PROF:  make -C lead-yolo_amd/csrc CXXFLAGS="... -DRC_PHASE_PROF" (touch ly_rf3c_bwd.hip first); rebuild without it afterwards.
The counters live in registers and perturb register allocation: read the SHARES, not the total."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import lead_yolo_amd as L
from lead_yolo_amd import capi
dev = torch.device("cuda:0")


def forward_profile():
    """python tools/rf3c_phase_prof.py fwd : phases of ly_rf3c_fwd_kernel<bf16, 2, 4, 2, false> (layer 17, eval forward bs=64), block 0 / thread 0"""
    model = B.build_model("s", dev)
    x = B.synth_batch(64, 640, 0, dev).to(torch.bfloat16)
    lib = capi.lib()
    lib.ly_rf3c_fwd_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
    with torch.no_grad():
        for _ in range(2):
            model(x)
        torch.cuda.synchronize()
        lib.ly_rf3c_fwd_prof(None, 1)
        model(x)
        torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    lib.ly_rf3c_fwd_prof(out, 0)
    v = list(out)[:5]
    tot = sum(v)
    print("ly_rf3c_fwd (layer 17 eval), one block: %d counter ticks" % tot)
    for n, q in zip(["prologue (plan, first loads, tables, weights)", "top barrier (the other waves' MFMAs)", "x chunk -> LDS, next chunk's loads, barrier",
                     "regenerate (VALU) + barrier", "contraction (MFMA)"], v):
        print("  %-48s %9d %5.1f %%" % (n, q, 100.0 * q / max(tot, 1)))

if len(sys.argv) > 1 and sys.argv[1] == "fwd":
    forward_profile()
    sys.exit(0)
model = B.build_model("s", dev, train=True)
opt = L.smart_optimizer(model, "SGD", 0.01, 0.937, 5e-4, fused=True)
cl = L.ComputeLoss(model)
imgs = B.synth_u8(64, 640, 0).to(dev); tg = B.synth_targets(64, 1).to(dev)
for _ in range(3):
    L.train_step(model, cl, opt, imgs, tg, amp=torch.bfloat16)
torch.cuda.synchronize()
lib = capi.lib()
lib.ly_rf3c_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.ly_rf3c_prof(None, 1)
L.train_step(model, cl, opt, imgs, tg, amp=torch.bfloat16)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
lib.ly_rf3c_prof(out, 0)
v = list(out)[:7]; tot = sum(v)
names = ["carries + top barrier", "barrier after staging", "MFMA phase + dcd tile write + barrier", "VALU pair loops (4 colours) + barrier", "dx output pass (C) / d_rfa reduction (A)", "commit (regs -> LDS)", "issue (next tile loads)"]
print("pass C, block 0 / thread 0, one launch: %d counter ticks" % tot)
for n, x in zip(names, v):
    print("  %-42s %10d  %5.1f %%" % (n, x, 100.0 * x / tot))
