import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import synth
from tests.test_gpu_modules import _bn_eps, _ctor, _load
from tests.test_gpu_backward import _oracle_grads, _hip_grads
from tests.test_oracle_golden import _run
kind, ctor, shape = "RFCBAMConv", (128, 128, 3, 2), (2, 128, 40, 40)
if len(sys.argv) > 1:
    ctor = tuple(int(v) for v in sys.argv[1].split(","))
    shape = tuple(int(v) for v in sys.argv[2].split(","))
torch.manual_seed(0)
m = _ctor(kind)(*ctor)
st = synth.synth_state(synth.shapes_of(m.state_dict()), 9100 + sum(shape) + len(kind))
_bn_eps(_load(m, st))
x = synth.synth_input(shape, 37 + shape[1])
with torch.no_grad():
    y0 = _run(kind, list(ctor), copy.deepcopy(st), x.clone(), True)[0]
r = synth.synth_input(tuple(y0.shape), 41 + shape[1])
yo, dxo, gpo = _oracle_grads(kind, ctor, st, x, r)
y, dx, gp = _hip_grads(m.to("cuda").train(), x, r)
def rel(a, b):
    a = a.detach().cpu().float(); b = b.detach().cpu().float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))
print("y", rel(y, yo), "dx", rel(dx, dxo))
for k, v in gpo.items():
    print(f"{k:28s} rel {rel(gp[k], v):.3e}  scale {float(v.abs().max()):.3e}")
e = (dx.cpu() - dxo).abs()
print("dx err by n:", e.amax((1, 2, 3)).tolist())
print("dx err by row:", [round(v, 3) for v in e.amax((0, 1, 3)).tolist()])
print("dx err by col:", [round(v, 3) for v in e.amax((0, 1, 2)).tolist()])
ec = e.amax((0, 2, 3)); print("worst channels:", ec.topk(5))
# ---- recompute check
from lead_yolo_amd import grad as GR
cap = {}
orig = GR.affine_backward
def spy(dy, u, a, b, act, mean, invstd, train):
    cap["v"] = (a.view(1, -1, 1, 1) * u + b.view(1, -1, 1, 1)).relu().clone()
    cap["dy"] = dy.clone()
    return orig(dy, u, a, b, act, mean, invstd, train)
GR.affine_backward = spy
m.zero_grad()
y2, dx2, gp2 = _hip_grads(m, x, r)
print("relu(a*u+b) vs y:", rel(cap["v"], y2), " dy vs r:", rel(cap["dy"], r))
ev = (cap["v"].cpu() - y2.cpu()).abs()
print("where:", (ev > 1e-3).nonzero()[:10].tolist())
gb_ = gp2["conv.1.bias"].cpu(); wb_ = gpo["conv.1.bias"]
d = (gb_ - wb_).abs()
print("conv.1.bias worst:", d.topk(6))
print("mine ", gb_[d.topk(6).indices].tolist()); print("oracle", wb_[d.topk(6).indices].tolist())
# direct: sum r*[y>0]
direct = (r * (yo > 0)).sum((0, 2, 3))
print("direct vs oracle", float((direct - wb_).abs().max()), "direct vs mine", float((direct - gb_).abs().max()))
