import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
cnt = [0]
def tap(name, t):
    if name == "rf.dy":
        cnt[0] += 1
    rec[(cnt[0], name)] = t.detach().clone()
GR.DEBUG_TAP = tap
runs = []
for it in range(4):
    m.zero_grad(); rec.clear(); cnt[0] = 0
    loss, _ = cl(m(x.cuda()), tg.cuda())
    loss.backward()
    runs.append(dict(rec))
for it in range(1, 4):
    print(it, {f"{k[0]}.{k[1][3:]}": f"{float((runs[it][k] - runs[0][k]).abs().max() / (runs[0][k].abs().max() + 1e-30)):.1e}" for k in sorted(runs[0]) if k[0] == 1})
