import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
_empty = torch.empty
def nan_empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float("nan"))
    return t
torch.empty = nan_empty
_empty_like = torch.empty_like
def nan_empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float("nan"))
    return t
torch.empty_like = nan_empty_like
import lead_yolo_amd as L
from oracle import synth
HW = int(sys.argv[1]) if len(sys.argv) > 1 else 320
cfg = L.load_cfg(scale="s")
torch.manual_seed(0)
m = L.Model(cfg)
st = synth.synth_state(synth.shapes_of(m.state_dict()), 4343)
st["model.23.anchors"] = m.model[-1].anchors.clone()
m.load_state_dict(st)
x = synth.synth_images(4, HW, 17).float() / 255
tg = synth.synth_targets(4, 18, per_image=4)
m = m.to("cuda").train()
cl = L.ComputeLoss(m)
rec = {}
def hook(key):
    def fn(mod, inp, out):
        if isinstance(out, torch.Tensor):
            rec[("y", key)] = bool(torch.isnan(out).any())
            out.register_hook(lambda g: rec.__setitem__(("dy", key), bool(torch.isnan(g).any())))
    return fn
for i, mod in enumerate(m.model):
    mod.register_forward_hook(hook(i))
loss, _ = cl(m(x.cuda()), tg.cuda())
loss.backward()
print("loss", float(loss))
print("nan in:", [k for k, v in rec.items() if v])
print("nan params:", [k for k, p in m.named_parameters() if torch.isnan(p.grad).any()][:20])
