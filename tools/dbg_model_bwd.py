import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
from oracle import functional as OF, synth
scale = sys.argv[1] if len(sys.argv) > 1 else "s"
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cfg = L.load_cfg(scale=scale)
torch.manual_seed(0)
m = L.Model(cfg)
st = synth.synth_state(synth.shapes_of(m.state_dict()), 4343)
st["model.23.anchors"] = m.model[-1].anchors.clone()
m.load_state_dict(st)
x = synth.synth_images(4, hw, 17).float() / 255
tg = synth.synth_targets(4, 18, per_image=4)
so = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and not k.endswith("anchors") else v.clone())
      for k, v in st.items()}
pred = OF.model_forward(so, cfg, x, m.stride, training=True)
for p in pred:
    p.retain_grad()
lo, _ = OF.compute_loss(pred, tg, m.model[-1].anchors, nc=1)
lo.backward()
m = m.to("cuda").train()
outs = m(x.cuda())
for o in outs:
    o.retain_grad()
loss, _ = L.ComputeLoss(m)(outs, tg.cuda())
loss.backward()
print("loss", float(loss), float(lo))
for i, (a, b) in enumerate(zip(outs, pred)):
    print(f"p{i} fwd rel {float((a.detach().cpu()-b.detach()).abs().max()/b.detach().abs().max()):.2e}  dpred rel {float((a.grad.cpu()-b.grad).abs().max()/b.grad.abs().max()):.2e}")
gscale = max(float(v.grad.abs().max()) for v in so.values() if v.requires_grad)
print("gscale", gscale)
for k, p in m.named_parameters():
    want = so[k].grad
    err = float((p.grad.cpu() - want).abs().max())
    sc = float(want.abs().max())
    flag = "  <<<" if err > 2e-3 * sc + 2e-5 * gscale else ""
    print(f"{k:45s} err {err:.3e} scale {sc:.3e} rel {err/(sc+1e-30):.2e}{flag}")
