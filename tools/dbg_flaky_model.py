import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
from lead_yolo_amd.modules import Lazy
from oracle import synth
cfg = L.load_cfg(scale="s")
torch.manual_seed(0)
m = L.Model(cfg)
st = synth.synth_state(synth.shapes_of(m.state_dict()), 4343)
st["model.23.anchors"] = m.model[-1].anchors.clone()
m.load_state_dict(st)
HW = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = synth.synth_images(4, HW, 17).float() / 255
tg = synth.synth_targets(4, 18, per_image=4)
m = m.to("cuda").train()
cl = L.ComputeLoss(m)
rec = {}
def hook(key):
    def fn(mod, inp, out):
        if isinstance(out, torch.Tensor):
            out.register_hook(lambda g: rec.__setitem__(key, g.detach().clone()))
    return fn
for i, mod in enumerate(m.model):
    mod.register_forward_hook(hook(i))
runs = []
for it in range(4):
    m.zero_grad(); rec.clear()
    loss, _ = cl(m(x.cuda()), tg.cuda())
    loss.backward()
    runs.append(({k: p.grad.clone() for k, p in m.named_parameters()}, dict(rec)))
g0, d0 = runs[0]
for it in range(1, 4):
    g, d = runs[it]
    bad = [(k, float((g[k] - g0[k]).abs().max() / (g0[k].abs().max() + 1e-30))) for k in g]
    bad = [b for b in bad if b[1] > 1e-4]
    badd = [(k, float((d[k] - d0[k]).abs().max() / (d0[k].abs().max() + 1e-30))) for k in d]
    badd = [b for b in badd if b[1] > 1e-4]
    print(it, "params off:", bad[:6], " layer-output grads off:", badd)
