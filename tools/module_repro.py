"""Which module's backward depends on what the caching allocator hands out?  Each module alone, bf16, train mode: forward + backward four times
from one state with the freed memory poisoned (0xFF bytes) in between; prints the relative difference of dx / parameter gradients between
consecutive runs.   python tools/module_repro.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L                                   # noqa: E402
from oracle import synth                                    # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
amp = None if "f32" in sys.argv else BF
cases = [("C3_CA", (88, 64, 1, False), (4, 88, 20, 20)), ("CA_Bottleneck", (32, 32, False, 1, 1.0), (4, 32, 20, 20)), ("CoordAtt", (32, 32), (4, 32, 20, 20)),
         ("Conv", (64, 64, 1, 1), (4, 64, 20, 20)), ("Conv", (32, 32, 3, 1), (4, 32, 20, 20)), ("RFCBAMConv", (64, 32, 1, 1), (4, 64, 10, 10)),
         ("BasicStage", (40, 1), (4, 40, 20, 20)), ("SPPF", (80, 80, 5), (4, 80, 5, 5)), ("PatchMerging_FasterNet", (40, 80, 2, 2), (4, 40, 20, 20))]
for kind, ctor, shape in cases:
    torch.manual_seed(0)
    m = getattr(L, kind)(*ctor)
    st = synth.synth_state(synth.shapes_of(m.state_dict()), 4400 + shape[1])
    m.load_state_dict(st)
    m = m.to(dev).train()
    x = synth.synth_input(shape, 5).to(dev)
    outs = []
    for it in range(4):
        if it >= 1:
            junk = [torch.full((n_,), -1, dtype=torch.int32, device=dev) for n_ in [1 << k for k in range(8, 22)] * 3]
            torch.cuda.synchronize()
            del junk
        m.load_state_dict({k: v.to(dev) for k, v in st.items()})
        for p in m.parameters():
            p.grad = None
        xt = x.clone().to(amp or torch.float32).requires_grad_(True)
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            y = m(xt)
        if it == 0:
            r = synth.synth_input(tuple(y.shape), 6).to(dev).to(y.dtype)
        y.backward(r)
        torch.cuda.synchronize()
        outs.append([y.detach().float().clone(), xt.grad.float().clone()] + [p.grad.float().clone() for p in m.parameters()])
    names = ["y", "dx"] + [n for n, _ in m.named_parameters()]
    worst = {}
    for i, nm in enumerate(names):
        d = max(float((outs[r_][i] - outs[r_ - 1][i]).norm()) for r_ in (1, 2, 3)) / (float(outs[0][i].norm()) + 1e-30)
        if d > 0:
            worst[nm] = d
    print(f"{kind}{ctor}: " + ("bit-identical over 4 runs" if not worst else " | ".join(f"{k} {v:.1e}" for k, v in sorted(worst.items(), key=lambda t: -t[1])[:6])), flush=True)
