"""GPU parity tests: the HIP modules (through the C-ABI) vs the oracle on identical inputs.
Tolerance from BASELINE.json north_star: within 1e-3 fp32."""
import copy

import numpy as np
import pytest
import torch

from oracle import functional as OF
from oracle import synth
from tests import golden_util as G

pytestmark = pytest.mark.gpu

ATOL = 1e-3
RTOL = 1e-3


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a CUDA/ROCm device"
    return torch.device("cuda:0")


def _load(mod, st):
    missing = mod.load_state_dict({k: v for k, v in st.items()}, strict=True)
    return mod


def _cmp(got, want, what):
    got = got.detach().float().cpu().numpy()
    want = want.detach().numpy() if isinstance(want, torch.Tensor) else want
    err = np.abs(got - want)
    tol = ATOL + RTOL * np.abs(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} elements off, max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
    return float(err.max())


@pytest.mark.parametrize("name", G.names("basicstage"))
def test_basicstage_golden(name):
    import lead_yolo_amd as L
    meta, arr = G.load(name)
    st = G.state_for(meta)
    x = synth.synth_input(meta["in_shape"], meta["seed"] + 1)
    m = _load(L.BasicStage(*meta["ctor"]), st).to(_dev()).eval()
    for mm in m.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.eps = 1e-3
    with torch.no_grad():
        y = m(x.to(_dev()))
    _cmp(y, arr["y_eval"], name)


@pytest.mark.parametrize("c,n,h,w", [(24, 3, 17, 13), (24, 2, 160, 160), (40, 2, 80, 80), (80, 2, 40, 40), (160, 3, 20, 20),
                                     (16, 1, 5, 7), (80, 1, 3, 200), (160, 2, 1, 1), (40, 1, 64, 2), (320, 1, 10, 12),
                                     (24, 12, 160, 160), (40, 24, 80, 80), (16, 8, 128, 128), (24, 7, 100, 240),       # >= 1024 patches: the persistent kernel
                                     (48, 2, 17, 13), (64, 1, 24, 20), (240, 1, 10, 12)])                             # other widths: the composed path
def test_basicstage_shapes_vs_oracle(c, n, h, w):
    """ragged tiles, tiles spanning images, 1-pixel-wide maps, real layer shapes"""
    import lead_yolo_amd as L
    torch.manual_seed(c * 1000 + h)
    m = L.BasicStage(c, 1)
    shapes = synth.shapes_of(m.state_dict())
    st = synth.synth_state(shapes, 5000 + c + h)
    _load(m, st)
    for mm in m.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.eps = 1e-3
    x = synth.synth_input((n, c, h, w), 77 + c)
    with torch.no_grad():
        want = OF.basic_stage(copy.deepcopy(st), "", x, False)
        got = m.to(_dev()).eval()(x.to(_dev()))
    _cmp(got, want, f"basicstage c={c} {n}x{h}x{w}")


def _bn_eps(m):
    for mm in m.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.eps, mm.momentum = 1e-3, 0.03
    return m


def _ctor(kind):
    import lead_yolo_amd as L
    return {"BasicStage": L.BasicStage, "PatchEmbed_FasterNet": L.PatchEmbed_FasterNet, "PatchMerging_FasterNet": L.PatchMerging_FasterNet,
            "RFCBAMConv": L.RFCBAMConv, "CoordAtt": L.CoordAtt, "CA_Bottleneck": L.CA_Bottleneck, "C3_CA": L.C3_CA, "SPPF": L.SPPF}[kind]


GOLDEN_MODULES = [n for pre in ("patch", "rfcbam", "coordatt", "cabottleneck", "c3ca", "sppf") for n in G.names(pre)]


@pytest.mark.parametrize("name", GOLDEN_MODULES)
def test_module_golden(name):
    """every hot-path module vs the vectors the reference itself produced"""
    meta, arr = G.load(name)
    st = G.state_for(meta)
    x = synth.synth_input(meta["in_shape"], meta["seed"] + 1)
    m = _bn_eps(_load(_ctor(meta["kind"])(*meta["ctor"]), st)).to(_dev()).eval()
    with torch.no_grad():
        y = m(x.to(_dev()))
    _cmp(y, arr["y_eval"], name)


def _oracle(kind, ctor, st, x):
    from tests.test_oracle_golden import _run
    with torch.no_grad():
        return _run(kind, ctor, copy.deepcopy(st), x, False)[0]


CASES = [
    ("PatchEmbed_FasterNet", (3, 24, 4, 4), (2, 3, 64, 96)),
    ("PatchEmbed_FasterNet", (3, 16, 4, 4), (1, 3, 32, 36)),
    ("PatchMerging_FasterNet", (24, 40, 2, 2), (2, 24, 40, 36)),
    ("PatchMerging_FasterNet", (160, 320, 2, 2), (1, 160, 10, 14)),
    ("RFCBAMConv", (160, 256, 1, 1), (3, 160, 20, 20)),
    ("RFCBAMConv", (256, 128, 1, 1), (2, 256, 40, 40)),
    ("RFCBAMConv", (128, 128, 3, 2), (2, 128, 80, 80)),
    ("RFCBAMConv", (256, 256, 3, 2), (2, 256, 40, 40)),
    ("RFCBAMConv", (64, 64, 3, 2), (1, 64, 21, 13)),
    ("RFCBAMConv", (32, 48, 3, 1), (2, 32, 9, 70)),
    ("RFCBAMConv", (512, 512, 3, 2), (1, 512, 12, 12)),
    ("C3_CA", (336, 256, 1, False), (2, 336, 40, 40)),
    ("C3_CA", (168, 128, 1, False), (1, 168, 80, 80)),
    ("C3_CA", (512, 512, 1, False), (2, 512, 20, 20)),
    ("C3_CA", (64, 64, 3, True), (2, 64, 13, 11)),
    ("C3_CA", (128, 96, 2, False), (1, 128, 7, 30)),
    ("CA_Bottleneck", (64, 64, True, 1, 1.0), (2, 64, 17, 9)),
    ("CoordAtt", (128, 128, 32), (2, 128, 11, 23)),
    ("SPPF", (160, 160, 5), (2, 160, 20, 20)),
]


@pytest.mark.parametrize("kind,ctor,shape", CASES)
def test_module_shapes_vs_oracle(kind, ctor, shape):
    """real layer shapes, ragged tiles, odd sizes, n>1 bottlenecks, shortcut"""
    torch.manual_seed(0)
    m = _ctor(kind)(*ctor)
    st = synth.synth_state(synth.shapes_of(m.state_dict()), 9000 + sum(shape) + len(kind))
    _bn_eps(_load(m, st))
    x = synth.synth_input(shape, 31 + shape[1])
    want = _oracle(kind, list(ctor), st, x)
    with torch.no_grad():
        got = m.to(_dev()).eval()(x.to(_dev()))
    _cmp(got, want, f"{kind}{ctor} {shape}")


def _cfg(scale):
    import lead_yolo_amd as L
    return L.load_cfg(scale=scale)


@pytest.mark.parametrize("scale", ["n", "s"])
@pytest.mark.parametrize("fuse_graph", [True, False])
def test_whole_model_golden(scale, fuse_graph):
    import lead_yolo_amd as L
    meta, arr = G.load(f"model_{scale}")
    pm, pa = G.load(f"parse_{scale}")
    st = G.state_for(meta, {"model.23.anchors": G.t(pa["anchors"])})
    m = L.Model(_cfg(scale), fuse_graph=fuse_graph)
    m.load_state_dict(st)
    m = m.to(_dev()).eval()
    hw = meta["hw"]
    x = synth.synth_images(2, max(hw), meta["seed"] + 1)[:, :, :hw[0], :hw[1]].float() / 255
    with torch.no_grad():
        z, outs = m(x.to(_dev()))
    _cmp(z, arr["z_eval"], f"model_{scale} z")
    for i, o in enumerate(outs):
        _cmp(o, arr[f"p{i}_eval"], f"model_{scale} p{i}")
    with torch.no_grad():
        zf = m.fuse()(x.to(_dev()))[0]
    _cmp(zf, arr["z_fused"], f"model_{scale} fused")


def test_whole_model_640_vs_oracle():
    """BASELINE config shape: lead-yolo-s, 640x640 (batch 2 to keep the CPU oracle to seconds)"""
    import lead_yolo_amd as L
    torch.manual_seed(0)
    m = L.Model(_cfg("s"))
    st = synth.synth_state(synth.shapes_of(m.state_dict()), 4242)
    st["model.23.anchors"] = m.model[-1].anchors.clone()
    m.load_state_dict(st)
    x = synth.synth_images(2, 640, 7).float() / 255
    with torch.no_grad():
        zo, outs_o = OF.model_forward(copy.deepcopy(st), _cfg("s"), x, m.stride, training=False)
        z, outs = m.to(_dev()).eval()(x.to(_dev()))
    assert z.shape == (2, 25200, 6)
    _cmp(z, zo, "model_s 640 z")
    for i, (a, b) in enumerate(zip(outs, outs_o)):
        _cmp(a, b, f"model_s 640 p{i}")


@pytest.mark.parametrize("scale,hw,bs", [("l", (64, 96), 2), ("n", (96, 64), 3), ("s", (1280, 1280), 1), ("l", (320, 320), 1),
                                         ("n", (640, 640), 1),          # BASELINE configs[0]: lead-yolo-n, one 640 x 640 image -> [1, 25200, 6]
                                         ("l", (1280, 1280), 1)])       # BASELINE configs[4]'s model AND input size, one image through the oracle
def test_other_scales_and_sizes_vs_oracle(scale, hw, bs):
    """lead-yolo-l (C up to 320 / 1024 channels, n=3 bottlenecks), lead-yolo-n, and the 1280x1280 input of
    BASELINE config 5, against the live oracle; configs[0]'s own shape (n @ 640, batch 1) on the HIP path"""
    import lead_yolo_amd as L
    torch.manual_seed(0)
    m = L.Model(_cfg(scale))
    st = synth.synth_state(synth.shapes_of(m.state_dict()), 6000 + hw[0])
    st["model.23.anchors"] = m.model[-1].anchors.clone()
    m.load_state_dict(st)
    x = synth.synth_images(bs, max(hw), 11)[:, :, :hw[0], :hw[1]].float() / 255
    with torch.no_grad():
        zo, outs_o = OF.model_forward(copy.deepcopy(st), _cfg(scale), x, m.stride, training=False)
        z, outs = m.to(_dev()).eval()(x.to(_dev()))
    if scale == "n" and hw == (640, 640):
        assert z.shape == (1, 25200, 6) and sum(p.numel() for p in m.parameters()) == 814382          # configs[0]'s plumbing check
    _cmp(z, zo, f"model_{scale} {hw} z")
    for i, (a, b) in enumerate(zip(outs, outs_o)):
        _cmp(a, b, f"model_{scale} {hw} p{i}")


TRAIN_MODULES = [n for n in (G.names("basicstage") + G.names("patch") + G.names("coordatt") + G.names("cabottleneck") + G.names("c3ca")
                              + G.names("sppf") + G.names("rfcbam"))]


@pytest.mark.parametrize("name", TRAIN_MODULES)
def test_module_train_forward_golden(name):
    """train-mode forward: batch-statistics BatchNorm output and the running-stat updates, vs the reference"""
    meta, arr = G.load(name)
    st = G.state_for(meta)
    x = synth.synth_input(meta["in_shape"], meta["seed"] + 1)
    m = _bn_eps(_load(_ctor(meta["kind"])(*meta["ctor"]), st)).to(_dev()).train()
    with torch.no_grad():
        y = m(x.to(_dev()))
    _cmp(y, arr["y_train"], name + " y_train")
    sd = m.state_dict()
    for k in arr:
        if k.startswith("post_"):
            _cmp(sd[k[5:]], arr[k], name + " " + k)
    nb = [v for k, v in sd.items() if k.endswith("num_batches_tracked")]
    assert all(int(v) == 1 for v in nb)


@pytest.mark.parametrize("scale", ["n", "s"])
def test_whole_model_train_forward_golden(scale):
    """train-mode forward of the whole detector (batch-statistics BN everywhere) vs the reference"""
    import lead_yolo_amd as L
    meta, arr = G.load(f"model_{scale}")
    pm, pa = G.load(f"parse_{scale}")
    st = G.state_for(meta, {"model.23.anchors": G.t(pa["anchors"])})
    m = L.Model(_cfg(scale))
    m.load_state_dict(st)
    m = m.to(_dev()).train()
    hw = meta["hw"]
    x = synth.synth_images(2, max(hw), meta["seed"] + 1)[:, :, :hw[0], :hw[1]].float() / 255
    with torch.no_grad():
        outs = m(x.to(_dev()))
    for i, o in enumerate(outs):
        got = o.detach().float().cpu().numpy()
        want = arr[f"p{i}_train"]
        err = np.abs(got - want)
        # P5 statistics come from 2x(2x3) pixels at this tiny input: allow a slightly wider band there
        assert err.max() <= 5e-3 + 5e-3 * np.abs(want).max(), (scale, i, err.max())


def test_graphed_forward_matches_eager():
    """serving mode: the forward captured into a hipGraph (with the SE / early-Detect branches forked onto the auxiliary
    stream) must return exactly what the eager single-stream forward returns, also after new input is copied in"""
    import lead_yolo_amd as L
    torch.manual_seed(0)
    m = L.Model(_cfg("s"))
    st = synth.synth_state(synth.shapes_of(m.state_dict()), 777)
    st["model.23.anchors"] = m.model[-1].anchors.clone()
    m.load_state_dict(st)
    m = m.to(_dev()).eval()
    for bs, parts in ((2, 1), (8, 2), (16, 4)):         # 1 stream + per-layer forks; 2 and 4 sub-batch streams writing batch slices
        x1 = (synth.synth_images(bs, 256, 5).float() / 255).to(_dev())
        x2 = (synth.synth_images(bs, 256, 6).float() / 255).to(_dev())
        g = L.GraphedForward(m, x1)
        assert g.parts == parts
        for x in (x1, x2, x1):
            with torch.no_grad():
                ze, pe = m(x)
            zg, pg = g(x)
            torch.cuda.synchronize()
            assert torch.equal(zg, ze)
            assert all(torch.equal(a, b) for a, b in zip(pg, pe))


def test_graphed_forward_matches_eager_bf16_full_batch():
    """configs[1]'s batch in bf16: the size-based kernel choice of RFCBAMConv k=3 (ops.rf3m_ok) must look at the WHOLE batch when the
    captured forward runs it as concurrent sub-batches — a sub-batch alone falls under the threshold, took the other kernel and the
    replay differed from the eager forward in the last bits (bench.py then fell back to eager launches: 1.12 -> 1.74 ms)"""
    import lead_yolo_amd as L
    from lead_yolo_amd import ops
    torch.manual_seed(0)
    m = L.Model(_cfg("s"))
    st = synth.synth_state(synth.shapes_of(m.state_dict()), 778)
    st["model.23.anchors"] = m.model[-1].anchors.clone()
    m.load_state_dict(st)
    m = m.to(_dev()).eval()
    x = (synth.synth_images(32, 640, 5).float() / 255).to(_dev()).to(torch.bfloat16)
    g = L.GraphedForward(m, x)
    assert g.parts == 2 and ops.CONCURRENT_PARTS == 1
    with torch.no_grad():
        ze, pe = m(x)
    zg, pg = g(x)
    torch.cuda.synchronize()
    assert torch.equal(zg, ze)
    assert all(torch.equal(a, b) for a, b in zip(pg, pe))


def test_rfcbam3_large_grid_vs_oracle():
    """RFCBAMConv k=3 at a grid large enough for the 256-channel tile's scalar-cache weight path (more than one block per CU,
    csrc/ly_rfcbam3.hip launch_rf3): images are independent in eval mode, the oracle checks the first two and the last."""
    kind, ctor, shape = "RFCBAMConv", (256, 256, 3, 2), (40, 256, 40, 40)      # 40 images x 7 row tiles = 280 blocks > 256 CUs
    torch.manual_seed(0)
    m = _ctor(kind)(*ctor)
    st = synth.synth_state(synth.shapes_of(m.state_dict()), 9100)
    _bn_eps(_load(m, st))
    x = synth.synth_input(shape, 77)
    pick = [0, 1, 39]
    want = _oracle(kind, list(ctor), st, x[pick])
    with torch.no_grad():
        got = m.to(_dev()).eval()(x.to(_dev()))
    _cmp(got[pick], want, "rfcbam3 256->256 s2, 40 images")


@pytest.mark.parametrize("name", G.names("rfcbam"))
def test_rfcbam_intermediates_golden(name):
    """the attention intermediates the fixtures hold (SE vector `ca`, the [max, mean] map, the receptive-field attention map),
    each against the reference's own value: a compensating error between the statistics pass and the main pass would show here"""
    from lead_yolo_amd import ops
    meta, arr = G.load(name)
    st = G.state_for(meta)
    x = synth.synth_input(meta["in_shape"], meta["seed"] + 1).to(_dev())
    m = _bn_eps(_load(_ctor(meta["kind"])(*meta["ctor"]), st)).to(_dev()).eval()
    xr, ld = ops.rows(ops.nhwc(x))
    n, c, h, w = xr.shape
    k, s = m.kernel_size, m.stride
    P = m._packed(2)
    with torch.no_grad():
        ca = m.se.attention(xr, ld, n, h * w, c)
        if k == 1:
            mm = ops.rfcbam_stats(xr, ld, n, h, w, c, 1, 1, a1=P["a1"], b1=P["b1"])
        else:
            ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
            th, tw = ops.pick_tile(ho, wo)
            mm = ops.rfcbam_stats(xr, ld, n, h, w, c, 3, s, wg=P["wq_stats"], th=th, tw=tw)
        rfa = ops.rfa_map(mm, P["w18"])
    _cmp(ca, arr["x_ca"], name + " ca")
    _cmp(mm.permute(0, 3, 1, 2), arr["x_mm"], name + " [max, mean] map")
    _cmp(rfa.unsqueeze(1), arr["x_rfa"], name + " rfa")
    # the fused launches the module itself uses: statistics + SE pooling partials in one pass, then SE linears + get_weight in one
    with torch.no_grad():
        if k == 1:
            mm2, part = ops.rfcbam_stats(xr, ld, n, h, w, c, 1, 1, a1=P["a1"], b1=P["b1"], gap=True)
        else:
            mm2, part = ops.rfcbam_stats(xr, ld, n, h, w, c, 3, s, wg=P["wq_stats"], th=th, tw=tw), ops.colsum(xr, ld, n, h * w, c)
        ca2, rfa2 = ops.rfcbam_mid(part, h * w, m.se.fc[0].weight.detach().contiguous(), m.se.fc[2].weight.detach().contiguous(), m.se.ratio, mm2, P["w18"])
    _cmp(ca2, arr["x_ca"], name + " ca (fused pass)")
    _cmp(mm2.permute(0, 3, 1, 2), arr["x_mm"], name + " [max, mean] map (fused pass)")
    _cmp(rfa2.unsqueeze(1), arr["x_rfa"], name + " rfa (fused pass)")


@pytest.mark.parametrize("name", G.names("coordatt"))
def test_coordatt_intermediates_golden(name):
    """CoordAtt's a_h / a_w gate factors against the reference's own values"""
    from lead_yolo_amd import ops
    meta, arr = G.load(name)
    st = G.state_for(meta)
    x = synth.synth_input(meta["in_shape"], meta["seed"] + 1).to(_dev())
    m = _bn_eps(_load(_ctor(meta["kind"])(*meta["ctor"]), st)).to(_dev()).eval()
    xr, ld = ops.rows(ops.nhwc(x))
    n, c, h, w = xr.shape
    with torch.no_grad():
        a_h, a_w = m.attention(xr, ld, n, h, w, c)
    _cmp(a_h.permute(0, 2, 1).unsqueeze(3), arr["x_a_h"], name + " a_h")
    _cmp(a_w.permute(0, 2, 1).unsqueeze(2), arr["x_a_w"], name + " a_w")


def test_full_batch_configs1_vs_oracle():
    """BASELINE configs[1] at its FULL batch (lead-yolo-s, bs=32, 640x640, eval): eager forward and hipGraph replay; the oracle
    checks two sampled images of every sub-batch of the serving-mode split (images are independent in eval mode)"""
    import lead_yolo_amd as L
    torch.manual_seed(0)
    m = L.Model(_cfg("s"))
    st = synth.synth_state(synth.shapes_of(m.state_dict()), 3232)
    st["model.23.anchors"] = m.model[-1].anchors.clone()
    m.load_state_dict(st)
    x = synth.synth_images(32, 640, 17).float() / 255
    pick = [0, 7, 8, 15, 16, 23, 24, 31]
    with torch.no_grad():
        zo, _ = OF.model_forward(copy.deepcopy(st), _cfg("s"), x[pick], m.stride, training=False)
        md = m.to(_dev()).eval()
        z, _ = md(x.to(_dev()))
        g = L.GraphedForward(md, x.to(_dev()))
        zg, _ = g()
        torch.cuda.synchronize()
    assert g.parts == 4
    _cmp(z[pick], zo, "configs[1] bs=32 eager")
    _cmp(zg[pick], zo, "configs[1] bs=32 graph replay")


def test_half_and_autocast_inputs():
    """SURVEY 8(b) call contract: fp16 inputs (`.half()` models) and autocast regions are accepted at the module edge: fp16 is
    computed by the bf16 kernels and handed back as fp16; fp32 under autocast(bf16) comes back as bf16"""
    import lead_yolo_amd as L
    torch.manual_seed(3)
    m = L.BasicStage(40, 1)
    st = synth.synth_state(synth.shapes_of(m.state_dict()), 40)
    _bn_eps(_load(m, st))
    x = synth.synth_input((2, 40, 16, 12), 9)
    with torch.no_grad():
        want = OF.basic_stage(copy.deepcopy(st), "", x, False)
        md = m.to(_dev()).eval()
        y16 = md(x.to(_dev()).half())
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yb = md(x.to(_dev()))
        with torch.autocast("cuda", dtype=torch.float16):
            yh = md(x.to(_dev()))
        y32 = md(x.to(_dev()))
    assert y16.dtype == torch.float16 and yb.dtype == torch.bfloat16 and yh.dtype == torch.float16 and y32.dtype == torch.float32
    scale = float(want.abs().max())
    for y in (y16, yb, yh):
        assert float((y.float().cpu() - want).abs().max()) <= 2 ** -6 * scale
    _cmp(y32, want, "fp32 path unchanged")


def test_conv_k3_s2_golden():
    """effective Conv with k = 3, stride 2 (models/common.py:1890-1910; not used by LEAD-YOLO.yaml): the stride-1 kernel + subsampling,
    against the reference's own vector"""
    import lead_yolo_amd as L
    meta, arr = G.load("conv_64_32_k3s2")
    st = G.state_for(meta)
    x = synth.synth_input(meta["in_shape"], meta["seed"] + 1)
    m = _bn_eps(_load(L.Conv(*meta["ctor"]), st)).to(_dev()).eval()
    with torch.no_grad():
        y = m(x.to(_dev()))
    _cmp(y, arr["y_eval"], "conv_64_32_k3s2")
    with pytest.raises(NotImplementedError):
        L.Conv(64, 32, 3, 1, g=64)                   # grouped / depthwise: not built, rejected at construction
    with pytest.raises(NotImplementedError):
        m.train()(x.to(_dev()))


def test_bf16x3_adversarial_bounds():
    """The fp32-storage path multiplies with a 3-term bf16 split (csrc/ly_tile.hpp: the lo*lo term, ~2^-16 relative per product, is
    dropped).  Worst cases for that error model, each against the fp32 oracle with the DOCUMENTED bound
        |err| <= 2^-14 * sum_k |w_k x_k|   (2^-16 per product with a 4x margin for accumulation order and the epilogue)
    which for well-conditioned sums is far inside the 1e-3 budget, and which is the honest bound when terms cancel:
      * K = 4608 (3x3 over 512 channels) with sign-alternating weights: massive cancellation, the result is ~0 against terms of O(1);
      * inputs spanning 2^-20 .. 2^20 in one tensor (the split is relative: small values keep their precision next to large ones);
      * BatchNorm running_var ~ 1e-6 (folded scale ~ 1e3 amplifies the contraction's absolute error)."""
    import lead_yolo_amd as L
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    # 1. 3x3 conv, 512 -> 64 channels, sign-alternating weights of equal magnitude, smooth positive input
    conv = L.Conv(512, 64, 3, 1)
    w = torch.ones(64, 512, 3, 3) * 0.05
    w.view(64, -1)[:, 1::2] *= -1
    w = w * (1 + 0.01 * torch.randn(w.shape, generator=g))
    x = 1.0 + 0.05 * torch.randn(2, 512, 12, 12, generator=g)
    with torch.no_grad():
        conv.conv.weight.copy_(w)
        conv.bn.weight.fill_(1.0); conv.bn.bias.zero_(); conv.bn.running_mean.zero_(); conv.bn.running_var.fill_(1.0)
        conv.bn.eps = 0.0
        conv.act = torch.nn.Identity()
        want = F.conv2d(x.double(), w.double(), padding=1).float()
        mag = F.conv2d(x.abs().double(), w.abs().double(), padding=1).float()              # sum_k |w_k x_k|
        got = conv.to(dev).eval()(x.to(dev)).float().cpu()
    assert float(want.abs().max()) < 0.2 * float(mag.max())                                # the case really cancels
    assert bool(((got - want).abs() <= 2.0 ** -14 * mag + 1e-7).all()), float(((got - want).abs() / mag).max())
    # 2. dynamic range: per-pixel scales 2^-20 .. 2^20 through a 1x1 conv
    pw = L.Conv(256, 128, 1, 1)
    scale = 2.0 ** torch.randint(-20, 21, (2, 1, 16, 16), generator=g).float()
    x2 = torch.randn(2, 256, 16, 16, generator=g) * scale
    with torch.no_grad():
        pw.bn.weight.fill_(1.0); pw.bn.bias.zero_(); pw.bn.running_mean.zero_(); pw.bn.running_var.fill_(1.0)
        pw.bn.eps = 0.0
        pw.act = torch.nn.Identity()
        w2 = pw.conv.weight.detach().clone()
        want2 = F.conv2d(x2.double(), w2.double()).float()
        mag2 = F.conv2d(x2.abs().double(), w2.abs().double()).float()
        got2 = pw.to(dev).eval()(x2.to(dev)).float().cpu()
    assert bool(((got2 - want2).abs() <= 2.0 ** -14 * mag2).all()), float(((got2 - want2).abs() / mag2).max())
    # 3. near-denormal BatchNorm variance: folded scale 1e3
    bn = L.Conv(128, 64, 1, 1)
    x3 = torch.randn(2, 128, 10, 10, generator=g)
    with torch.no_grad():
        bn.bn.running_var.fill_(1e-6); bn.bn.eps = 0.0; bn.bn.running_mean.normal_(0, 1e-3, generator=g)
        bn.act = torch.nn.Identity()
        w3 = bn.conv.weight.detach().clone()
        s3 = (bn.bn.weight / torch.sqrt(bn.bn.running_var)).detach()
        u = F.conv2d(x3.double(), w3.double())
        want3 = ((u - bn.bn.running_mean.double().view(1, -1, 1, 1)) * s3.double().view(1, -1, 1, 1) + bn.bn.bias.double().view(1, -1, 1, 1)).float()
        mag3 = F.conv2d(x3.abs().double(), w3.abs().double()).float() * s3.abs().view(1, -1, 1, 1)
        got3 = bn.to(dev).eval()(x3.to(dev)).float().cpu()
    assert bool(((got3 - want3).abs() <= 2.0 ** -14 * mag3 + 1e-6 * want3.abs()).all()), float(((got3 - want3).abs() / mag3).max())


def test_patchembed_uint8_image_matches_float_path():
    """a uint8 batch given to the patch embedding (pixel / 255 folded into the gather, train.py:309) == the fp32 path on x / 255"""
    import lead_yolo_amd as L
    dev = _dev()
    torch.manual_seed(3)
    m = L.PatchEmbed_FasterNet(3, 24, 4, 4).to(dev).eval()
    with torch.no_grad():
        m.norm.running_mean.normal_(0, 0.1)
        m.norm.running_var.uniform_(0.5, 1.5)
    x8 = torch.randint(0, 256, (2, 3, 64, 96), dtype=torch.uint8, device=dev)
    with torch.no_grad():
        ref = m(x8.float() / 255)
        got = m(x8)
    assert got.dtype == torch.float32
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("idt", [torch.bfloat16, torch.float16])
def test_patchembed_half_image_is_gathered_as_it_is(idt, monkeypatch):
    """the `im.half()` batch of a reduced-precision forward (val.py:207): the patch gather reads the 16-bit NCHW image directly
    (LY_GATHER_PATCH_NCHW_BF16 / _F16) — same bits as gathering its fp32 copy, and no fp32 copy is made"""
    import lead_yolo_amd as L
    from lead_yolo_amd import ops
    dev = _dev()
    torch.manual_seed(4)
    m = L.PatchEmbed_FasterNet(3, 40, 4, 4).to(dev).eval()
    with torch.no_grad():
        m.norm.running_mean.normal_(0, 0.1)
        m.norm.running_var.uniform_(0.5, 1.5)
    x = torch.rand(3, 3, 72, 100, device=dev).to(idt)
    seen = []
    real = ops.gemm
    monkeypatch.setattr(ops, "gemm", lambda **kw: (seen.append(kw["a0"].dtype), real(**kw))[1])
    with torch.no_grad():
        got = m(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ref = m(x.float())                                    # fp32 image, bf16 output: the route taken before
    assert seen == [idt, torch.float32]
    assert got.dtype == idt and got.shape == (3, 40, 18, 25)
    assert torch.equal(got.float(), ref.to(idt).float())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_detect_level_one_launch_matches_gemm_plus_tail(dt):
    """ly_detect_level (head 1x1 convolution + decode in one launch, models/yolo.py:88-120) vs the GEMM + ly_detect_tail pair it replaces:
    same z / raw maps up to the bf16 rounding of the pair's intermediate buffer"""
    import lead_yolo_amd as L
    from lead_yolo_amd import modules
    dev = _dev()
    torch.manual_seed(5)
    ch = (64, 128, 256) if dt == torch.float32 else (128, 256, 512)
    det = L.Detect(nc=1, anchors=((10, 13, 16, 30, 33, 23), (30, 61, 62, 45, 59, 119), (116, 90, 156, 198, 373, 326)), ch=ch).to(dev).eval()
    det.stride = torch.tensor([8., 16., 32.], device=dev)
    det.anchors /= det.stride.view(-1, 1, 1)
    xs = [torch.randn(3, c, s, s + 1, device=dev).to(dt).contiguous(memory_format=torch.channels_last) for c, s in zip(ch, (20, 10, 5))]
    with torch.no_grad():
        z1, p1 = det([t for t in xs])
        modules.FUSED_DETECT_LEVEL = False
        try:
            z0, p0 = det([t for t in xs])
        finally:
            modules.FUSED_DETECT_LEVEL = True
    tol = dict(rtol=2e-2, atol=2e-2) if dt == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    assert z1.shape == z0.shape == (3, 3 * (20 * 21 + 10 * 11 + 5 * 6), 6)
    for a, b in zip(p1, p0):
        torch.testing.assert_close(a, b, **tol)
    torch.testing.assert_close(z1, z0, rtol=tol["rtol"], atol=tol["atol"] * 40)      # (xy / wh are scaled by stride and anchors)


def test_train_step_uint8_equals_float_batch():
    """forward_backward on the uint8 batch (no fp32 copy of the images) reproduces the gradients of the float / 255 batch"""
    import lead_yolo_amd as L
    from lead_yolo_amd.train import forward_backward
    dev = _dev()
    torch.manual_seed(0)
    imgs = torch.randint(0, 256, (2, 3, 64, 64), dtype=torch.uint8, device=dev)
    tg = torch.tensor([[0, 0, 0.5, 0.5, 0.3, 0.3], [1, 0, 0.4, 0.6, 0.2, 0.5]], device=dev)
    grads = []
    for as_u8 in (True, False):
        torch.manual_seed(1)
        model = L.Model(L.load_cfg(scale="n")).to(dev).train()
        loss_fn = L.ComputeLoss(model)
        assert model.u8_input
        x = imgs if as_u8 else imgs.float() / 255
        forward_backward(model, loss_fn, x, tg)
        grads.append(model.model[0].proj.weight.grad.clone())
    # two runs of the SAME batch already differ by ~1e-3 relative (float atomics in the batch statistics, then ReLU kinks): direction + norm
    a, b = grads[0].flatten().double(), grads[1].flatten().double()
    assert torch.dot(a, b) / (a.norm() * b.norm()) > 0.9999
    assert (a - b).norm() <= 1e-2 * b.norm()
