"""Layer-wise backward check inside the whole model: for every layer of the HIP model capture (input, grad of output),
replay that layer alone through the oracle with the SAME input and cotangent, compare dx and parameter gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import lead_yolo_amd as L
from lead_yolo_amd.modules import Lazy
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
m = m.to("cuda").train()
rec = {}


def mat(v):
    if isinstance(v, Lazy):
        return v.materialize()
    return v


def hook(i):
    def fn(mod, inp, out):
        xin = inp[0]
        r = dict(kind=type(mod).__name__)
        if isinstance(xin, (list, tuple)):
            return
        xt = mat(xin)
        r["x"] = xt.detach().clone()
        if isinstance(xin, torch.Tensor) and xin.requires_grad:
            xin.register_hook(lambda g: r.__setitem__("dx", g.detach().clone()))
        if isinstance(out, torch.Tensor):
            out.register_hook(lambda g: r.__setitem__("dy", g.detach().clone()))
            r["y"] = out.detach().clone()
        rec[i] = r
    return fn


for i, mod in enumerate(m.model):
    if isinstance(mod, torch.nn.Sequential) and not isinstance(mod, (L.BasicStage,)):
        for j, sub in enumerate(mod):
            sub.register_forward_hook(hook(f"{i}.{j}"))
    else:
        mod.register_forward_hook(hook(str(i)))
outs = m(x.cuda())
loss, _ = L.ComputeLoss(m)(outs, tg.cuda())
loss.backward()
gp = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}

layers, save = OF.parse_graph(cfg, 3)
kinds = {str(Lr["i"]): (Lr["kind"], Lr["args"], Lr["repeated"]) for Lr in layers}


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


for key, r in rec.items():
    base = key.split(".")[0]
    kind, args, rep = kinds[base]
    if "dy" not in r or kind in ("nn.Upsample", "Concat", "Detect"):
        continue
    pfx = f"model.{key}."
    so = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in st.items()
          if k.startswith(pfx)}
    xt = r["x"].cpu().contiguous().clone().requires_grad_(True)
    if kind == "PatchEmbed_FasterNet":
        y = OF.patch_conv(so, pfx, xt, args[2], "proj", True)
    elif kind == "PatchMerging_FasterNet":
        y = OF.patch_conv(so, pfx, xt, args[2], "reduction", True)
    elif kind == "BasicStage":
        y = OF.basic_stage(so, pfx, xt, True)
    elif kind == "SPPF":
        y = OF.sppf(so, pfx, xt, args[2], True)
    elif kind == "RFCBAMConv":
        y = OF.rfcbam(so, pfx, xt, args[2], args[3], True)
    elif kind == "C3_CA":
        y = OF.c3_ca(so, pfx, xt, args[3], True)
    else:
        continue
    y.backward(r["dy"].cpu())
    line = f"{key:6s} {kind:24s} y {rel(r['y'].cpu(), y.detach()):.1e}"
    if "dx" in r:
        line += f"  dx {rel(r['dx'].cpu(), xt.grad):.1e}"
    worst = max(((rel(gp[k], v.grad), k) for k, v in so.items() if v.requires_grad and float(v.grad.abs().max()) > 1e-6), default=(0, ""))
    line += f"  worst param {worst[0]:.1e} ({worst[1][len(pfx):]})"
    print(line)
