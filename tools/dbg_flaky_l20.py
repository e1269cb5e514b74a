import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import lead_yolo_amd as L
from lead_yolo_amd import grad as GR
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
def tap(t, name):
    v = t.view_as(t)
    v.register_hook(lambda g: rec.__setitem__(name, g.detach().clone()))
    return v
def rf_dbg(mod, x):
    tagn = f"L{mod.i}"
    n, c, h, w = x.shape
    xa, xb = tap(x, tagn + ".dx_main"), tap(x, tagn + ".dx_se")
    gap = GR.PoolHW.apply(xb)[:, :h].mean(1)
    ca = torch.sigmoid(F.linear(F.relu(F.linear(gap, mod.se.fc[0].weight)), mod.se.fc[2].weight))
    ca = tap(ca, tagn + ".d_ca")
    g, cv = mod.generate, mod.conv
    out = GR.RfcbamFn.apply(mod, xa, ca, g[0].weight, g[1].weight, g[1].bias, mod.get_weight[0].weight, cv[0].weight, cv[0].bias,
                            cv[1].weight, cv[1].bias)
    rec[tagn + ".y"] = out.detach().clone()
    return tap(out, tagn + ".dy")
GR.rfcbam_train = rf_dbg
runs = []
for it in range(4):
    m.zero_grad(); rec.clear()
    loss, _ = cl(m(x.cuda()), tg.cuda())
    loss.backward()
    runs.append(dict(rec))
for it in range(1, 4):
    print(it, {k: f"{float((runs[it][k] - runs[0][k]).abs().max() / (runs[0][k].abs().max() + 1e-30)):.1e}" for k in sorted(runs[0])})
