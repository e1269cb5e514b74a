"""Pins oracle/functional.py against golden vectors emitted by the reference itself
(oracle/gen_golden.py).  CPU only."""
import copy

import numpy as np
import pytest
import torch

from oracle import functional as OF
from oracle import synth
from tests import golden_util as G

TOL = dict(rtol=2e-5, atol=2e-5)


def _run(kind, ctor, st, x, training, want_extras=False):
    p = ""
    if kind == "BasicStage":
        return OF.basic_stage(st, p, x, training), None
    if kind == "PatchEmbed_FasterNet":
        return OF.patch_conv(st, p, x, ctor[2], "proj", training), None
    if kind == "PatchMerging_FasterNet":
        return OF.patch_conv(st, p, x, ctor[2], "reduction", training), None
    if kind == "RFCBAMConv":
        y, ex = OF.rfcbam(st, p, x, ctor[2], ctor[3], training, return_intermediates=True)
        return y, ex
    if kind == "CoordAtt":
        y, ex = OF.coord_att(st, p, x, training, return_intermediates=True)
        return y, ex
    if kind == "CA_Bottleneck":
        return OF.ca_bottleneck(st, p, x, ctor[2], training), None
    if kind == "C3_CA":
        return OF.c3_ca(st, p, x, ctor[3], training), None
    if kind == "SPPF":
        return OF.sppf(st, p, x, ctor[2], training), None
    if kind == "Conv":
        return OF.conv_bn_silu(st, p, x, ctor[2], ctor[3], training), None
    raise NotImplementedError(kind)


MODULE_CASES = [n for pre in ("basicstage", "patch", "rfcbam", "coordatt", "cabottleneck", "c3ca", "sppf", "conv_")
                for n in G.names(pre)]


@pytest.mark.parametrize("name", MODULE_CASES)
def test_module_matches_reference(name):
    meta, arr = G.load(name)
    st = G.state_for(meta)
    x = synth.synth_input(meta["in_shape"], meta["seed"] + 1)
    kind, ctor = meta["kind"], meta["ctor"]
    with torch.no_grad():
        y, ex = _run(kind, ctor, copy.deepcopy(st), x.clone(), False)
    np.testing.assert_allclose(y.numpy(), arr["y_eval"], **TOL)
    if ex:
        for k, v in ex.items():
            np.testing.assert_allclose(v.numpy().reshape(arr["x_" + k].shape), arr["x_" + k], **TOL)
    # train mode: batch-stat BN, running-stat update, input gradient
    stt = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
           for k, v in st.items()}
    xt = x.clone().requires_grad_(True)
    yt, _ = _run(kind, ctor, stt, xt, True)
    np.testing.assert_allclose(yt.detach().numpy(), arr["y_train"], rtol=1e-4, atol=1e-4)
    r = synth.synth_input(yt.shape, meta["seed"] + 2)
    (yt * r).sum().backward()
    np.testing.assert_allclose(xt.grad.numpy(), arr["dx_train"], rtol=2e-4, atol=2e-4)
    for k, gn in meta["grad_norms"].items():
        got = float(stt[k].grad.double().norm())
        # a bias feeding a train-mode BN has an exactly-zero true gradient: only rounding noise remains
        assert abs(got - gn) <= 1e-3 * gn + 2e-4, (k, got, gn)
    for k in arr:
        if k.startswith("post_"):
            np.testing.assert_allclose(stt[k[5:]].detach().numpy(), arr[k], rtol=1e-5, atol=1e-6)


def _cfg(scale):
    import yaml, os
    with open(os.path.join(os.path.dirname(__file__), "..", "lead-yolo_amd", "cfg", "LEAD-YOLO.yaml")) as f:
        cfg = yaml.safe_load(f)
    gd, gw = {"n": (0.33, 0.25), "s": (0.33, 0.50), "l": (1.0, 1.0)}[scale]
    cfg["depth_multiple"], cfg["width_multiple"] = gd, gw
    return cfg


@pytest.mark.parametrize("scale", ["n", "s", "l"])
def test_parse_graph_table(scale):
    meta, arr = G.load(f"parse_{scale}")
    layers, save = OF.parse_graph(_cfg(scale))
    assert save == meta["save"]
    assert [L["kind"].split(".")[-1] for L in layers] == [r["type"] for r in meta["rows"]]
    assert [L["f"] for L in layers] == [r["f"] for r in meta["rows"]]


@pytest.mark.parametrize("scale", ["n", "s"])
def test_whole_model_forward(scale):
    meta, arr = G.load(f"model_{scale}")
    pm, pa = G.load(f"parse_{scale}")
    st = G.state_for(meta, {"model.23.anchors": G.t(pa["anchors"])})
    hw = meta["hw"]
    x = synth.synth_images(2, max(hw), meta["seed"] + 1)[:, :, :hw[0], :hw[1]].float() / 255
    stride = G.t(pa["stride"])
    with torch.no_grad():
        z, outs = OF.model_forward(copy.deepcopy(st), _cfg(scale), x, stride, training=False)
    np.testing.assert_allclose(z.numpy(), arr["z_eval"], rtol=1e-4, atol=1e-4)
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.numpy(), arr[f"p{i}_eval"], rtol=1e-4, atol=1e-4)
    with torch.no_grad():
        pt = OF.model_forward(copy.deepcopy(st), _cfg(scale), x, stride, training=True)
    for i, o in enumerate(pt):
        np.testing.assert_allclose(o.numpy(), arr[f"p{i}_train"], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("case", ["rand", "edge", "empty"])
def test_build_targets_and_loss(case):
    meta, arr = G.load("loss_n")
    preds = [G.t(arr[f"pred{i}"]).requires_grad_(True) for i in range(3)]
    anchors = G.t(arr["anchors"])
    tg = G.t(arr[f"{case}_targets"])
    tcls, tbox, indices, anch = OF.build_targets([p.shape for p in preds], tg, anchors, meta["hyp"]["anchor_t"])
    for i in range(3):
        got = np.stack([v.numpy() for v in indices[i]]).astype(np.int64)
        assert got.dtype == np.int64 and np.array_equal(got, arr[f"{case}_idx{i}"])          # bit-exact
        assert np.array_equal(tcls[i].numpy().astype(np.int64), arr[f"{case}_tcls{i}"])
        np.testing.assert_array_equal(tbox[i].numpy(), arr[f"{case}_tbox{i}"])
        np.testing.assert_array_equal(anch[i].numpy(), arr[f"{case}_anch{i}"])
    loss, items = OF.compute_loss(preds, tg, anchors, nc=1, hyp=meta["hyp"])
    np.testing.assert_allclose(loss.detach().numpy(), arr[f"{case}_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(items.numpy(), arr[f"{case}_items"], rtol=1e-5, atol=1e-6)
    loss.backward()
    for i in range(3):
        np.testing.assert_allclose(preds[i].grad.numpy(), arr[f"{case}_dpred{i}"], rtol=1e-4, atol=1e-6)


def test_three_sgd_steps():
    meta, arr = G.load("trainsteps_n")
    pm, pa = G.load("parse_n")
    st = G.state_for(meta, {"model.23.anchors": G.t(pa["anchors"])})
    cfg = _cfg("n")
    stride, anchors = G.t(pa["stride"]), G.t(pa["anchors"])
    params = {k: v for k, v in st.items() if v.is_floating_point() and "running" not in k and not k.endswith("anchors")}
    groups = OF.param_groups(list(st))
    groups = {g: [k for k in ks if k in params] for g, ks in groups.items()}
    assert len(groups["decay"]) == meta["groups"]["n_decay"]
    assert len(groups["bn"]) == meta["groups"]["n_bn"]
    assert len(groups["bias"]) == meta["groups"]["n_bias"]
    bufs = {}
    B = meta["B"]
    for step in range(3):
        imgs = synth.synth_images(B, 64, 910 + step).float() / 255
        tg = synth.synth_targets(B, 920 + step, per_image=3)
        for p in params.values():
            p.requires_grad_(True)
            p.grad = None
        pred = OF.model_forward(st, cfg, imgs, stride, training=True)
        loss, items = OF.compute_loss(pred, tg, anchors, nc=1)
        loss.backward()
        assert abs(loss.item() - meta["losses"][step]) <= 2e-4 * abs(meta["losses"][step]) + 1e-4
        grads = {k: p.grad for k, p in params.items()}
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
        coef = torch.clamp(10.0 / (total + 1e-6), max=1.0)
        grads = {k: g * coef for k, g in grads.items()}
        with torch.no_grad():
            for gname, wd in (("decay", meta["weight_decay"]), ("bn", 0.0), ("bias", 0.0)):
                sub = {k: params[k] for k in groups[gname]}
                OF.sgd_nesterov_step(sub, grads, bufs, meta["lr0"], meta["momentum"], wd)
    for k in meta["probe"]:
        np.testing.assert_allclose(st[k].detach().numpy(), arr["final_" + k], rtol=2e-4, atol=2e-5)


def test_metrics_oracle_matches_reference_vectors():
    """oracle/metrics.py (val.py:79-101 process_batch, utils/metrics.py:31-123 ap_per_class / compute_ap, :406-424 box_iou) against what the
    unmodified reference returned for the same detections / labels (tests/golden/metrics_cases.npz, oracle/gen_golden.py metrics)"""
    from oracle import metrics as OM
    meta, arr = G.load("metrics_cases")
    for k, case in enumerate(meta["cases"]):
        det, lab = arr[f"det{k}"], arr[f"lab{k}"]
        np.testing.assert_array_equal(OM.box_iou(lab[:, 1:], det[:, :4]), arr[f"iou{k}"])
        correct = OM.process_batch(det, lab)
        np.testing.assert_array_equal(correct, arr[f"correct{k}"])
        if case["nl"]:
            got = OM.ap_per_class(correct, det[:, 4], det[:, 5], lab[:, 0])
            for name, v in zip(("tp", "fp", "p", "r", "f1", "ap", "cls"), got):
                np.testing.assert_allclose(np.asarray(v, np.float64), arr[f"{name}{k}"], rtol=0, atol=1e-12, err_msg=f"case {k} {name}")
    # val.py:183-188 on two "images"
    stats = [(OM.process_batch(arr[f"det{k}"], arr[f"lab{k}"]), arr[f"det{k}"][:, 4], arr[f"det{k}"][:, 5], arr[f"lab{k}"][:, 0]) for k in (0, 4)]
    mp, mr, map50, m = OM.mean_results(stats)
    assert 0 < m <= map50 <= 1 and 0 < mp <= 1 and 0 < mr <= 1
