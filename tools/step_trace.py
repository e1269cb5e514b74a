"""Which launch first produces different bits between two runs of the same training step?  Records the output of every gemm / conv3x3 /
up2_bwd / bnact_bwd_apply / coordatt / pool call of forward + backward (clones), run twice with the allocator's free memory poisoned in
between, and prints the first records that differ.   python tools/step_trace.py [bf16|f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L                                   # noqa: E402
from lead_yolo_amd import ops                               # noqa: E402
from oracle import synth                                    # noqa: E402

dev = torch.device("cuda:0")
amp = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] != "f32") else None
torch.manual_seed(0)
m = L.Model(L.load_cfg(scale="n"))
st = synth.synth_state(synth.shapes_of(m.state_dict()), 7373)
st["model.23.anchors"] = m.model[-1].anchors.clone()
m.load_state_dict(st)
m = m.to(dev).train()
cl = L.ComputeLoss(m)
imgs = synth.synth_images(4, 160, 71).to(dev)
tg = synth.synth_targets(4, 72, per_image=4).to(dev)
bufs0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
REC = []


def wrap(name, outkey=None, ret=False):
    fn = getattr(ops, name)

    def w(*a, **k):
        r = fn(*a, **k)
        t = r if ret else k.get(outkey)
        if isinstance(t, (tuple, list)):
            t = t[0]
        if torch.is_tensor(t):
            torch.cuda.synchronize()
            REC[-1].append((name, tuple(t.shape), t.detach().float().clone()))
        return r
    setattr(ops, name, w)


for nm, key in (("gemm", "out"), ("conv3x3", "out")):
    wrap(nm, outkey=key)
for nm in ("up2_bwd", "coordatt_mlp_bwd", "coordatt_gate", "pool_hw", "sum_rows", "se_bwd"):
    if hasattr(ops, nm):
        wrap(nm, ret=True)
orig_apply = ops.bnact_bwd_apply


def apply_rec(*a, **k):
    r = orig_apply(*a, **k)
    torch.cuda.synchronize()
    REC[-1].append(("bnact_bwd_apply", tuple(a[12].shape), a[12].detach().float().clone()))
    return r


ops.bnact_bwd_apply = apply_rec
for it in range(3):
    if it >= 1:
        junk = [torch.full((n_,), -1, dtype=torch.int32, device=dev) for n_ in [1 << k for k in range(8, 24)] * 3]
        torch.cuda.synchronize()
        del junk
    m.load_state_dict(bufs0)
    for p in m.parameters():
        p.grad = None
    REC.append([])
    with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
        pred = m(imgs.float() / 255)
        loss, _ = cl(pred, tg)
    nf = len(REC[-1])
    loss.backward()
    torch.cuda.synchronize()
    print(f"run {it}: {nf} forward records, {len(REC[-1]) - nf} backward records")
for a, b in ((0, 1), (1, 2)):
    shown = 0
    for i, (ra, rb) in enumerate(zip(REC[a], REC[b])):
        if ra[0] != rb[0] or ra[1] != rb[1]:
            print(f"runs {a}/{b}: record {i} differs in KIND/shape: {ra[:2]} vs {rb[:2]}")
            break
        d = float((ra[2] - rb[2]).norm()) / (float(ra[2].norm()) + 1e-30)
        if d > 0:
            print(f"runs {a}/{b}: record {i} ({'fwd' if i < nf else 'bwd'} #{i - nf if i >= nf else i}) {ra[0]} {ra[1]} differs by {d:.2e}")
            shown += 1
            if shown >= 6:
                break
    if not shown:
        print(f"runs {a}/{b}: every record identical")
