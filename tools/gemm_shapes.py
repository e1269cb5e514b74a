"""Per-call table of the 1x1 / patch GEMM launches (forward, statistics and data-gradient contractions) of ONE eager training step of
lead-yolo-s: shape, gather / prologue, kernel time (HIP events on the launch stream), algorithmic GB/s.  Dev tool.
    python tools/gemm_shapes.py [bf16|f32] [bs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import lead_yolo_amd as L
from lead_yolo_amd import ops

dev = torch.device("cuda:0")
amp = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else None
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
model = B.build_model("s", dev, train=True)
opt = L.smart_optimizer(model, "SGD", 0.01, 0.937, 5e-4, fused=True)
cl = L.ComputeLoss(model)
imgs = B.synth_u8(bs, 640, 0).to(dev)
tg = B.synth_targets(bs, 1).to(dev)
for _ in range(3):
    L.train_step(model, cl, opt, imgs, tg, amp=amp)
torch.cuda.synchronize()
LOG = []
_g = ops.gemm


def gemm(**q):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _g(**q)
    e1.record()
    es = 2 if (q["out"] if q.get("out") is not None else q["a0"]).dtype == torch.bfloat16 else 4
    nb = es * q["M"] * (q["K"] + (q["N"] if q.get("out") is not None else 0))
    LOG.append((f"M={q['M']:7d} K={q['K']:5d} N={q['N']:4d} gather={q.get('gather', 0)} pro={q.get('pro', 0)} "
                f"{'stats ' if q.get('stats') is not None else ''}{'nostore' if q.get('out') is None else ''}", nb, e0, e1))


ops.gemm = gemm
L.train_step(model, cl, opt, imgs, tg, amp=amp)
torch.cuda.synchronize()
tot = 0.0
for label, nb, e0, e1 in LOG:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    print(f"{us:8.1f} us  {nb / us / 1e3:7.0f} GB/s  {label}")
print(f"{len(LOG)} gemm launches, {tot:.1f} us (event time includes the launch gaps of the eager step)")
