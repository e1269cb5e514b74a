"""Generate tests/golden/*.npz by running the UNMODIFIED reference (imported from /root/reference in
the build container) on synthetic weights/inputs from oracle/synth.py.  TEST INFRASTRUCTURE.

    python -m oracle.gen_golden            # regenerates every fixture

A fixture holds: json meta (kind, ctor args, seeds, key->shape table, state checksum) + expected
output arrays produced by the reference modules themselves.  No reference source is stored.
"""
import copy
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_import, synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def _save(name, meta, arrays):
    os.makedirs(OUT, exist_ok=True)
    arrays = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
                        **arrays)
    sz = os.path.getsize(os.path.join(OUT, name + ".npz"))
    print(f"  {name}: {sz / 1024:.1f} KiB")


def _prep(ref, mod):
    """What DetectionModel.__init__ does to every module: BN eps/momentum (utils/torch_utils.py:212-221)."""
    ref.torch_utils.initialize_weights(mod)
    return mod


def _load_synth(mod, seed):
    shapes = synth.shapes_of(mod.state_dict())
    st = synth.synth_state(shapes, seed)
    for k, v in mod.state_dict().items():
        if st.get(k) is None:
            st[k] = v.clone()
    mod.load_state_dict(st)
    return shapes, synth.checksum(st)


def module_case(ref, name, kind, ctor, in_shape, seed, factory, extras=None):
    torch.manual_seed(0)
    mod = _prep(ref, factory(*ctor))
    shapes, csum = _load_synth(mod, seed)
    x = synth.synth_input(in_shape, seed + 1)
    r_seed = seed + 2
    arrays = {}
    # eval
    mod.eval()
    with torch.no_grad():
        y = mod(x.clone())
    arrays["y_eval"] = _np(y)
    if extras:
        for k, v in extras(mod, x).items():
            arrays["x_" + k] = _np(v)
    # train (fresh copy so running stats start from the synthetic values)
    tm = copy.deepcopy(mod).train()
    xt = x.clone().requires_grad_(True)
    yt = tm(xt)
    r = synth.synth_input(yt.shape, r_seed)
    (yt * r).sum().backward()
    arrays["y_train"] = _np(yt)
    arrays["dx_train"] = _np(xt.grad)
    gn = {k: float(p.grad.double().norm()) for k, p in tm.named_parameters()}
    post = {k: _np(v) for k, v in tm.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    for k, v in post.items():
        arrays["post_" + k] = v
    meta = dict(kind=kind, ctor=list(ctor), in_shape=list(in_shape), seed=seed, shapes=shapes, checksum=csum,
                grad_norms=gn)
    _save(name, meta, arrays)


def gen_modules(ref):
    C, R = ref.common, ref.rfa
    print("modules:")
    for c in (16, 24, 40, 80, 160):
        module_case(ref, f"basicstage_c{c}", "BasicStage", (c, 1), (2, c, 12, 10), 100 + c, C.BasicStage)
    module_case(ref, "patchembed_3_24", "PatchEmbed_FasterNet", (3, 24, 4, 4), (2, 3, 32, 24), 201, C.PatchEmbed_FasterNet)
    module_case(ref, "patchmerge_24_40", "PatchMerging_FasterNet", (24, 40, 2, 2), (2, 24, 12, 10), 202,
                C.PatchMerging_FasterNet)
    module_case(ref, "patchmerge_80_160", "PatchMerging_FasterNet", (80, 160, 2, 2), (2, 80, 8, 6), 203,
                C.PatchMerging_FasterNet)

    def rf_extras(mod, x):
        # intermediates recomputed with the reference's own sub-modules (models/rfa.py:113-127)
        from einops import rearrange
        with torch.no_grad():
            b, c = x.shape[:2]
            k = mod.kernel_size
            ca = mod.se(x)
            g = mod.generate(x)
            h, w = g.shape[2:]
            G = rearrange(g.view(b, c, k * k, h, w), 'b c (n1 n2) h w -> b c (h n1) (w n2)', n1=k, n2=k)
            mx, _ = torch.max(G, dim=1, keepdim=True)
            mm = torch.cat((mx, torch.mean(G, dim=1, keepdim=True)), 1)
            return {"ca": ca.view(b, c), "mm": mm, "rfa": mod.get_weight(mm)}

    for (ci, co, k, s, hw, sd) in ((160, 256, 1, 1, (6, 7), 301), (256, 128, 1, 1, (8, 6), 302),
                                   (128, 128, 3, 2, (8, 10), 303), (256, 256, 3, 2, (6, 8), 304),
                                   (64, 64, 3, 2, (9, 11), 305), (80, 128, 1, 1, (5, 5), 306)):
        module_case(ref, f"rfcbam_{ci}_{co}_k{k}s{s}", "RFCBAMConv", (ci, co, k, s), (2, ci, *hw), sd, R.RFCBAMConv,
                    extras=rf_extras)

    def ca_extras(mod, x):
        with torch.no_grad():
            n, c, h, w = x.shape
            xh = mod.pool_h(x)
            xw = mod.pool_w(x).permute(0, 1, 3, 2)
            y = mod.act(mod.bn1(mod.conv1(torch.cat([xh, xw], dim=2))))
            yh, yw = torch.split(y, [h, w], dim=2)
            return {"a_h": mod.conv_h(yh).sigmoid(), "a_w": mod.conv_w(yw.permute(0, 1, 3, 2)).sigmoid()}

    module_case(ref, "coordatt_64", "CoordAtt", (64, 64, 32), (2, 64, 7, 9), 401, C.CoordAtt, extras=ca_extras)
    module_case(ref, "coordatt_256", "CoordAtt", (256, 256, 32), (2, 256, 5, 6), 402, C.CoordAtt, extras=ca_extras)
    module_case(ref, "cabottleneck_64", "CA_Bottleneck", (64, 64, False, 1, 1.0), (2, 64, 7, 9), 403, C.CA_Bottleneck)
    module_case(ref, "cabottleneck_32_sc", "CA_Bottleneck", (32, 32, True, 1, 1.0), (2, 32, 6, 5), 404, C.CA_Bottleneck)
    for (c1, c2, n, sc, hw, sd) in ((336, 256, 1, False, (6, 7), 501), (168, 128, 1, False, (8, 10), 502),
                                    (256, 256, 1, False, (6, 8), 503), (512, 512, 1, False, (5, 4), 504),
                                    (64, 64, 3, True, (7, 6), 505), (88, 64, 2, False, (6, 6), 506)):
        module_case(ref, f"c3ca_{c1}_{c2}_n{n}{'s' if sc else ''}", "C3_CA", (c1, c2, n, sc), (2, c1, *hw), sd, C.C3_CA)
    module_case(ref, "sppf_160", "SPPF", (160, 160, 5), (2, 160, 7, 9), 601, C.SPPF)
    module_case(ref, "conv_64_32_k3s2", "Conv", (64, 32, 3, 2), (2, 64, 8, 10), 602, C.Conv)


def _cfg(ref, scale):
    import yaml
    with open(os.path.join(ref_import.REFERENCE_ROOT, "models", "LEAD-YOLO.yaml"), encoding="ascii", errors="ignore") as f:
        cfg = yaml.safe_load(f)
    gd, gw = {"n": (0.33, 0.25), "s": (0.33, 0.50), "l": (1.0, 1.0)}[scale]
    cfg["depth_multiple"], cfg["width_multiple"] = gd, gw
    return cfg


def gen_parse(ref):
    print("parse tables:")
    for scale in ("n", "s", "l"):
        torch.manual_seed(0)
        m = ref.yolo.Model(_cfg(ref, scale))
        rows = []
        for mod in m.model:
            rows.append(dict(i=mod.i, f=mod.f, type=mod.type.split(".")[-1], np=int(mod.np)))
        det = m.model[-1]
        meta = dict(scale=scale, rows=rows, save=list(m.save), nparams=int(sum(p.numel() for p in m.parameters())),
                    shapes=synth.shapes_of(m.state_dict()))
        arrays = dict(stride=_np(m.stride), anchors=_np(det.anchors),
                      det_bias=np.concatenate([_np(c.bias) for c in det.m]))
        # fused-model bookkeeping (models/yolo.py:213-233)
        fm = copy.deepcopy(m).fuse()
        meta["fused_nparams"] = int(sum(p.numel() for p in fm.parameters()))
        meta["fused_keys"] = list(fm.state_dict().keys())
        _save(f"parse_{scale}", meta, arrays)


def gen_model(ref):
    print("whole model:")
    for scale, hw, seed in (("n", (64, 64), 701), ("s", (64, 96), 702)):
        torch.manual_seed(0)
        m = ref.yolo.Model(_cfg(ref, scale))
        shapes, csum = _load_synth(m, seed)        # anchors / stride buffers keep the model's own values
        x = synth.synth_images(2, 0, seed + 1)[:, :, :0]  # placeholder to keep RNG recipe simple
        x = (synth.synth_images(2, max(hw), seed + 1)[:, :, :hw[0], :hw[1]].float() / 255)
        arrays = {}
        m.eval()
        with torch.no_grad():
            z, outs = m(x.clone())
        arrays["z_eval"] = _np(z)
        for i, o in enumerate(outs):
            arrays[f"p{i}_eval"] = _np(o)
        # per-layer activations (eval) for bisecting: mean/abs-mean + small strided sample
        feats = {}
        hooks = [mod.register_forward_hook(lambda mod_, inp, out, i=mod.i: feats.__setitem__(i, out)) for mod in m.model[:-1]]
        with torch.no_grad():
            m(x.clone())
        for h in hooks:
            h.remove()
        for i, f in feats.items():
            arrays[f"feat{i}_stats"] = np.array([float(f.double().mean()), float(f.double().abs().mean())])
        fm = copy.deepcopy(m).fuse().eval()
        with torch.no_grad():
            arrays["z_fused"] = _np(fm(x.clone())[0])
        tm = copy.deepcopy(m).train()
        pt = tm(x.clone())
        for i, o in enumerate(pt):
            arrays[f"p{i}_train"] = _np(o)
        meta = dict(scale=scale, hw=list(hw), seed=seed, shapes=shapes, checksum=csum, batch=2)
        _save(f"model_{scale}", meta, arrays)


def gen_loss(ref):
    print("loss / build_targets:")
    torch.manual_seed(0)
    m = ref.yolo.Model(_cfg(ref, "n"))
    import yaml
    with open(os.path.join(ref_import.REFERENCE_ROOT, "data", "hyps", "hyp.scratch-low.yaml")) as f:
        hyp = yaml.safe_load(f)
    m.hyp = hyp
    m.nc = 1
    det = m.model[-1]
    cl = ref.loss.ComputeLoss(m)
    B, S = 3, 128
    cases = {}
    t = synth.synth_targets(B, 801, per_image=6)
    # edge cases: box on a grid boundary, centre within 0.5 of the border, tiny and huge boxes
    edge = torch.tensor([[0, 0, 0.5, 0.5, 0.1, 0.1], [1, 0, 0.0625, 0.0625, 0.05, 0.3], [2, 0, 0.999, 0.999, 0.2, 0.2],
                         [0, 0, 0.002, 0.5, 0.01, 0.01], [1, 0, 0.25, 0.75, 0.9, 0.9], [2, 0, 0.5 - 1e-4, 0.5 + 1e-4, 0.07, 0.11]])
    cases["rand"] = t
    cases["edge"] = edge
    cases["empty"] = torch.zeros(0, 6)
    arrays, meta = {}, dict(B=B, S=S, hyp={k: float(v) for k, v in hyp.items()}, cases=list(cases))
    preds = [synth.synth_input((B, 3, S // s, S // s, 6), 810 + i) for i, s in enumerate((8, 16, 32))]
    for i, p in enumerate(preds):
        arrays[f"pred{i}"] = _np(p)
    arrays["anchors"] = _np(det.anchors)
    for name, tg in cases.items():
        arrays[f"{name}_targets"] = _np(tg)
        tcls, tbox, indices, anch = cl.build_targets(preds, tg)
        for i in range(3):
            b, a, gj, gi = indices[i]
            arrays[f"{name}_idx{i}"] = np.stack([_np(b), _np(a), _np(gj), _np(gi)]).astype(np.int64)
            arrays[f"{name}_tbox{i}"] = _np(tbox[i])
            arrays[f"{name}_anch{i}"] = _np(anch[i])
            arrays[f"{name}_tcls{i}"] = _np(tcls[i]).astype(np.int64)
        ps = [p.clone().requires_grad_(True) for p in preds]
        loss, items = cl(ps, tg)
        loss.backward()
        arrays[f"{name}_loss"] = _np(loss)
        arrays[f"{name}_items"] = _np(items)
        for i in range(3):
            arrays[f"{name}_dpred{i}"] = _np(ps[i].grad)
    _save("loss_n", meta, arrays)


def gen_loss_multiclass(ref):
    """the class-BCE branch of ComputeLoss (utils/loss.py:168-173: nc > 1) with label smoothing and both positive weights off their
    defaults, from the reference's own ComputeLoss on a lead-yolo-n head built with nc = 3"""
    print("loss, nc = 3:")
    torch.manual_seed(0)
    cfg = _cfg(ref, "n")
    cfg["nc"] = 3
    m = ref.yolo.Model(cfg, nc=3)
    import yaml
    with open(os.path.join(ref_import.REFERENCE_ROOT, "data", "hyps", "hyp.scratch-low.yaml")) as f:
        hyp = yaml.safe_load(f)
    hyp.update(label_smoothing=0.1, cls_pw=1.3, obj_pw=0.8)
    m.hyp = hyp
    det = m.model[-1]
    assert det.nc == 3 and det.no == 8
    cl = ref.loss.ComputeLoss(m)
    B, S = 3, 128
    tg = synth.synth_targets(B, 811, per_image=6)
    tg[:, 1] = torch.randint(0, 3, (tg.shape[0],), generator=torch.Generator().manual_seed(812)).float()
    arrays, meta = {}, dict(B=B, S=S, nc=3, hyp={k: float(v) for k, v in hyp.items()})
    preds = [synth.synth_input((B, 3, S // s, S // s, 8), 820 + i) for i, s in enumerate((8, 16, 32))]
    for i, p in enumerate(preds):
        arrays[f"pred{i}"] = _np(p)
    arrays["anchors"] = _np(det.anchors)
    arrays["targets"] = _np(tg)
    tcls, tbox, indices, anch = cl.build_targets(preds, tg)
    for i in range(3):
        arrays[f"tcls{i}"] = _np(tcls[i]).astype(np.int64)
    ps = [p.clone().requires_grad_(True) for p in preds]
    loss, items = cl(ps, tg)
    loss.backward()
    arrays["loss"] = _np(loss)
    arrays["items"] = _np(items)
    for i in range(3):
        arrays[f"dpred{i}"] = _np(ps[i].grad)
    _save("loss_nc3", meta, arrays)


def gen_trainsteps(ref):
    """T: 3 optimiser steps of lead-yolo-n at 64x64 exactly as train.py:295-341 does them on CPU fp32
    (amp off => scaler is a no-op; nbs=64, batch 4 => accumulate=16 in train.py, here we step every
    iteration to exercise the update; weight_decay scaled by bs*accumulate/nbs with accumulate=1)."""
    print("train steps:")
    torch.manual_seed(0)
    m = ref.yolo.Model(_cfg(ref, "n"))
    shapes, csum = _load_synth(m, 901)
    import yaml
    with open(os.path.join(ref_import.REFERENCE_ROOT, "data", "hyps", "hyp.scratch-low.yaml")) as f:
        hyp = yaml.safe_load(f)
    m.hyp, m.nc = hyp, 1
    B = 4
    wd = hyp["weight_decay"] * B * 1 / 64
    opt = ref.torch_utils.smart_optimizer(m, "SGD", hyp["lr0"], hyp["momentum"], wd)
    cl = ref.loss.ComputeLoss(m)
    m.train()
    arrays, losses = {}, []
    for step in range(3):
        imgs = synth.synth_images(B, 64, 910 + step).float() / 255
        tg = synth.synth_targets(B, 920 + step, per_image=3)
        pred = m(imgs)
        loss, items = cl(pred, tg)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=10.0)
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
        arrays[f"items{step}"] = _np(items)
    sd = m.state_dict()
    probe = ["model.0.proj.weight", "model.1.blocks.0.mlp.0.weight", "model.1.blocks.0.mlp.1.running_var",
             "model.9.conv.0.bias", "model.12.m.0.ca.conv_h.bias", "model.23.m.0.bias"]
    for k in probe:
        arrays["final_" + k] = _np(sd[k])
    meta = dict(scale="n", B=B, seed=901, shapes=shapes, checksum=csum, losses=losses, lr0=hyp["lr0"],
                momentum=hyp["momentum"], weight_decay=wd, probe=probe,
                groups={"n_bias": len(opt.param_groups[0]["params"]), "n_decay": len(opt.param_groups[1]["params"]),
                        "n_bn": len(opt.param_groups[2]["params"])})   # order: utils/torch_utils.py:337-342
    _save("trainsteps_n", meta, arrays)


def gen_metrics(ref):
    """val.py's metric chain on synthetic detections / labels: `utils.metrics.box_iou`, `val.process_batch` and `utils.metrics.ap_per_class`
    of the unmodified reference (pins oracle/metrics.py)"""
    import importlib
    print("metrics:")
    val = importlib.import_module("val")
    rng = np.random.default_rng(20240)
    arrays, cases = {}, []
    for k, (nd, nl, nc) in enumerate([(40, 7, 1), (100, 20, 3), (5, 1, 1), (60, 12, 2), (300, 33, 1), (3, 0, 1)]):
        lab = np.zeros((nl, 5), np.float32)
        lab[:, 0] = rng.integers(0, nc, nl)
        xy, wh = rng.uniform(0, 200, (nl, 2)), rng.uniform(10, 80, (nl, 2))
        lab[:, 1:3], lab[:, 3:5] = xy, xy + wh
        det = np.zeros((nd, 6), np.float32)
        if nl:
            src = rng.integers(0, nl, nd)
            det[:, :4] = lab[src, 1:] + rng.normal(0, 6, (nd, 4))
            det[:, 5] = np.where(rng.uniform(size=nd) < 0.8, lab[src, 0], rng.integers(0, nc, nd))
        else:
            det[:, :4] = rng.uniform(0, 200, (nd, 4))
        det[:, 4] = rng.uniform(0.01, 1, nd)
        iou = ref.metrics.box_iou(torch.from_numpy(lab[:, 1:]), torch.from_numpy(det[:, :4])).numpy()
        correct = val.process_batch(torch.from_numpy(det), torch.from_numpy(lab), torch.linspace(0.5, 0.95, 10)).numpy()
        arrays.update({f"det{k}": det, f"lab{k}": lab, f"iou{k}": iou, f"correct{k}": correct})
        if nl:
            out = ref.metrics.ap_per_class(correct, det[:, 4], det[:, 5], lab[:, 0], plot=False, names={i: str(i) for i in range(nc)})
            for name, v in zip(("tp", "fp", "p", "r", "f1", "ap", "cls"), out):
                arrays[f"{name}{k}"] = np.asarray(v, np.float64)
        cases.append(dict(nd=nd, nl=nl, nc=nc))
    _save("metrics_cases", dict(cases=cases), arrays)


def main():
    import sys
    ref = ref_import.load()
    torch.set_num_threads(8)
    gens = dict(modules=gen_modules, parse=gen_parse, model=gen_model, loss=gen_loss, loss_multiclass=gen_loss_multiclass,
                trainsteps=gen_trainsteps, metrics=gen_metrics)
    for name in (sys.argv[1:] or list(gens)):            # `python oracle/gen_golden.py loss_multiclass` regenerates one group
        gens[name](ref)


if __name__ == "__main__":
    main()
