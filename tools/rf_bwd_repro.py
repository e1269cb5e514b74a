"""Run-to-run stability of the RFCBAMConv k=3 backward (bf16): the same module step N times from one state, per-tensor differences between
runs, for the recompute backward (csrc/ly_rf3c_bwd.hip, grad.RC_BWD = True) and the streamed one.   python tools/rf_bwd_repro.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L                                   # noqa: E402
from lead_yolo_amd import grad                              # noqa: E402
from oracle import synth                                    # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
names = ("conv.0.weight", "generate.0.weight", "generate.1.weight", "generate.1.bias", "conv.1.weight", "conv.1.bias", "se.fc.0.weight", "se.fc.2.weight",
         "get_weight.0.weight")
for ctor, shape in (((64, 64, 3, 2), (4, 64, 48, 48)), ((128, 128, 3, 2), (4, 128, 40, 40))):
    c, o, k, s = ctor
    torch.manual_seed(0)
    m = L.RFCBAMConv(*ctor)
    st = synth.synth_state(synth.shapes_of(m.state_dict()), 9100)
    m.load_state_dict(st)
    for mm in m.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.eps, mm.momentum = 1e-3, 0.03
    m = m.to(dev).train()
    x = synth.synth_input(shape, 77).to(dev).to(BF)
    r = synth.synth_input((shape[0], o, shape[2] // s, shape[3] // s), 78).to(dev).to(BF)
    for rc in (True, False):
        grad.RC_BWD = rc
        outs = []
        for _ in range(6):
            for p in m.parameters():
                p.grad = None
            xt = x.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=BF):
                y = m(xt)
            y.backward(r)
            torch.cuda.synchronize()
            outs.append([y.detach().float().clone(), xt.grad.float().clone()] + [dict(m.named_parameters())[n].grad.float().clone() for n in names])
        print(f"C={c} O={o} {'recompute (ly_rf3c_bwd)' if rc else 'streamed'} backward, 6 runs: ", end="")
        rep = []
        for i, nm in enumerate(("y", "dx") + names):
            base = outs[0][i]
            d = max(float((o_[i] - base).abs().max()) for o_ in outs[1:])
            rep.append(f"{nm} {d / (float(base.abs().max()) + 1e-30):.1e}")
        print(" | ".join(rep), flush=True)
grad.RC_BWD = True
