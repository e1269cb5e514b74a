"""Eight-wave / tap-split passes B, C of the RFCBAMConv k=3 recompute backward (csrc/ly_rf3c_bwd8.hip) against the four-wave kernels on the same
module and inputs (every gradient), and per-kernel times at the layer-17 shape:  python tools/rf3c_bwd8_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L  # noqa: E402
from lead_yolo_amd import grad, ops  # noqa: E402

dev = torch.device("cuda:0")


def run(rc8, shape, ci, co, prof=False):
    grad.RC8_BWD = rc8
    torch.manual_seed(0)
    m = L.RFCBAMConv(ci, co, 3, 2).to(dev).train()
    x = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    torch.manual_seed(1)
    y = m(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    out = dict(dx=x.grad.float().clone(), **{k: p.grad.float().clone() for k, p in m.named_parameters()})
    if prof:
        for p in m.parameters():
            p.grad = None
        x.grad = None
        ops.PROFILE = []
        y = m(x)
        y.backward(dy)
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
        for name, _, _, a, b, _ in recs:
            if "rf3c_bwd" in name or "wgrad" in name:
                print(f"      {a.elapsed_time(b) * 1e3:8.1f} us  {name}")
    return out


for shape, ci, co in [((2, 64, 20, 24), 64, 64), ((3, 128, 18, 30), 128, 128), ((64, 128, 80, 80), 128, 128)]:
    big = shape[0] == 64
    a = run(False, shape, ci, co, prof=big)
    b = run(True, shape, ci, co, prof=big)
    worst = 0.0
    for k in a:
        d = (a[k] - b[k]).norm() / (a[k].norm() + 1e-30)
        worst = max(worst, float(d))
        print(f"  {shape} {k:28s} rel L2 diff {float(d):.3e}  |ref| {float(a[k].norm()):.3e}")
    print(f"{shape}: worst relative difference {worst:.3e}")
