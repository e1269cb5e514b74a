"""Run-to-run reproducibility of one training step (forward + loss + backward) of lead-yolo-n from one state: per-parameter relative
difference of the gradients between runs, in model order (where does the noise enter?).   python tools/step_repro.py [bf16|f32] [scale=s] [bs=64] [size=640] [sink]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L                                   # noqa: E402
from oracle import synth                                    # noqa: E402

dev = torch.device("cuda:0")
amp = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] != "f32") else None
torch.manual_seed(0)
SCALE = next((a.split("=")[1] for a in sys.argv if a.startswith("scale=")), "n")
BS = int(next((a.split("=")[1] for a in sys.argv if a.startswith("bs=")), "4"))
SIZE = int(next((a.split("=")[1] for a in sys.argv if a.startswith("size=")), "160"))
m = L.Model(L.load_cfg(scale=SCALE))
st = synth.synth_state(synth.shapes_of(m.state_dict()), 7373)
st["model.23.anchors"] = m.model[-1].anchors.clone()
m.load_state_dict(st)
m = m.to(dev).train()
cl = L.ComputeLoss(m)
imgs = synth.synth_images(BS, SIZE, 71).to(dev)
tg = synth.synth_targets(BS, 72, per_image=4).to(dev)
SINK = "sink" in sys.argv          # gradients through optim.FusedSGD's gradient sink (what train_step uses): one warm-up step creates the storage
if SINK:
    opt = L.smart_optimizer(m, "SGD", 0.01, 0.937, 5e-4, fused=True)
    L.train_step(m, cl, opt, imgs, tg, amp=amp)
    st0 = synth.synth_state(synth.shapes_of(m.state_dict()), 7373)
    st0["model.23.anchors"] = m.model[-1].anchors.clone()
    m.load_state_dict(st0)
bufs0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
runs = []
hooks, acts = [], []
for i, layer in enumerate(m.model):
    def hk(mod, gin, gout, i=i):
        g = gout[0] if isinstance(gout, (tuple, list)) else gout
        if torch.is_tensor(g):
            acts[-1][i] = g.detach().float().clone()
    hooks.append(layer.register_full_backward_hook(hk))
poison = "--poison" in sys.argv
for it in range(4):
    if poison and it >= 1:
        # hunt for reads of uninitialised memory: fill what the caching allocator will hand out next with 0xFF bytes (NaN as fp32 and as bf16)
        junk = [torch.full((n_,), -1, dtype=torch.int32, device=dev) for n_ in [1 << k for k in range(8, 24)] * 3]
        torch.cuda.synchronize()
        del junk
    m.load_state_dict(bufs0)
    for p in m.parameters():
        if SINK and p.grad is not None:
            p.grad.zero_()
        else:
            p.grad = None
    acts.append({})
    with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
        pred = m(imgs.float() / 255)
        loss, _ = cl(pred, tg)
    loss.backward()
    torch.cuda.synchronize()
    runs.append({n: p.grad.detach().float().clone() for n, p in m.named_parameters()})
g0 = runs[0]
tot = float(torch.cat([g.flatten() for g in g0.values()]).norm())
print("gradient w.r.t. each layer's OUTPUT, relative difference between runs (backward order):")
for i in sorted(acts[0], reverse=True):
    a = acts[0][i]
    d = [float((acts[r][i] - acts[r - 1][i]).norm()) / (float(a.norm()) + 1e-30) for r in (1, 2, 3)]
    print(f"  layer {i:2d} {type(m.model[i]).__name__:24s} d(out) rel diff run1-run0 {d[0]:.2e}  run2-run1 {d[1]:.2e}  run3-run2 {d[2]:.2e}")
print("parameter gradients with the largest relative difference:")
rows = sorted(((max(float((runs[r][n] - g0[n]).norm()) for r in (1, 2)) / max(float(g0[n].norm()), 1e-3 * tot), n) for n in g0), reverse=True)
for d, n in rows[:12]:
    print(f"  {n:48s} {d:.2e}")
nz = [(d, n) for d, n in rows if d > 0]
print(f"{len(nz)} of {len(rows)} parameter gradients differ between runs at all:")
for d, n in nz:
    print(f"  {n:48s} {d:.2e}  {tuple(g0[n].shape)}")
