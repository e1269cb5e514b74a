"""Per-call table of the weight-gradient launches of ONE eager bf16/f32 training step of lead-yolo-s bs=64: shape, gather, kernel time
(HIP events on the launch stream), algorithmic GB/s and TFLOP/s.  Dev tool.
    python tools/wgrad_shapes.py [bf16|f32] [bs]"""
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
_w, _g = ops.wgrad, ops.wgrad_group


def desc(q):
    g = "rows" if (q.get("ks", 1) == 1 and q.get("stride", 1) == 1 and not q.get("nchw") and not q.get("up2")) else \
        f"k{q.get('ks', 1)}s{q.get('stride', 1)}{'u' if q.get('up2') else ''}{'n' if q.get('nchw') else ''}"
    return f"M={q['M']:7d} N={q['N']:4d} K={q.get('ks', 1) ** 2 * q['Cin']:5d} {g}{' pro' if q.get('x_scale') is not None else ''}"


def timed(fn, label, flops, nbytes):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    LOG.append((label, flops, nbytes, e0, e1))


def wgrad(_now=False, **q):
    if not _now and ops._wgrad_deferrable(q):
        return _w(**q)                                   # waits in ops._WgradQueue: timed when its group leaves
    es = q["x"].element_size()
    k = q.get("ks", 1) ** 2 * q["Cin"]
    timed(lambda: _w(_now=_now, **q), desc(q), 2.0 * q["M"] * q["N"] * k, es * q["M"] * (q["N"] + q["Cin"]) + 4.0 * q["N"] * k)


def wgrad_group(problems, _now=False):
    if not _now and all(ops._wgrad_deferrable(q) for q in problems):
        return _g(problems)
    if not (2 <= len(problems) <= 4) or not all(ops._wgrad_groupable(q, problems[0]["x"].dtype) for q in problems):
        for q in problems:                               # not one launch: its members appear as their own lines
            wgrad(_now=_now, **q)
        return
    es = problems[0]["x"].element_size()
    timed(lambda: _g(problems, _now=_now), " + ".join(desc(q) for q in problems), sum(2.0 * q["M"] * q["N"] * q["Cin"] for q in problems),
          sum(es * q["M"] * (q["N"] + q["Cin"]) + 4.0 * q["N"] * q["Cin"] for q in problems))


ops.wgrad, ops.wgrad_group = wgrad, wgrad_group
import lead_yolo_amd.grad as G
for mod in (G,):
    for name in ("wgrad", "wgrad_group"):
        if hasattr(mod, name):
            setattr(mod, name, globals()[name])
L.train_step(model, cl, opt, imgs, tg, amp=amp)
torch.cuda.synchronize()
tot = 0.0
for label, flops, nbytes, e0, e1 in LOG:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    print(f"{us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s  {flops / us / 1e6:7.1f} TFLOP/s  {label}")
print(f"{len(LOG)} wgrad launches, {tot:.1f} us (event time includes launch gaps of the eager step)")
