"""Host-side logic of the product (no GPU, no compute): yaml rules, channel table, state_dict
layout, stride/anchor bookkeeping (bit-exact), fuse() key layout, C-ABI export table."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import lead_yolo_amd as L
from tests import golden_util as G


@pytest.mark.parametrize("scale", ["n", "s", "l"])
def test_parse_table_matches_reference(scale):
    meta, arr = G.load(f"parse_{scale}")
    torch.manual_seed(0)
    m = L.Model(L.load_cfg(scale=scale))
    assert m.save == meta["save"]
    rows = [dict(i=mod.i, f=mod.f, type=mod.type.split(".")[-1], np=int(mod.np)) for mod in m.model]
    assert rows == meta["rows"]
    assert sum(p.numel() for p in m.parameters()) == meta["nparams"]
    ours = [(k, list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in m.state_dict().items()]
    assert ours == [tuple(s) if False else (s[0], s[1], s[2]) for s in meta["shapes"]]
    # integer / index bookkeeping is bit-exact
    assert np.array_equal(m.stride.numpy(), arr["stride"])
    assert np.array_equal(m.model[-1].anchors.numpy(), arr["anchors"])
    got_bias = np.concatenate([c.bias.detach().numpy() for c in m.model[-1].m])
    # Detect biases: random init + deterministic offsets; compare the offsets' effect on shape only
    assert got_bias.shape == arr["det_bias"].shape
    fm = m.fuse()
    assert list(fm.state_dict().keys()) == meta["fused_keys"]
    assert sum(p.numel() for p in fm.parameters()) == meta["fused_nparams"]


def test_bn_policy_and_make_divisible():
    m = L.Model(L.load_cfg(scale="n"))
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            assert mod.eps == 1e-3 and mod.momentum == 0.03
    assert L.make_divisible(40 * 0.5, 8) == 24 and L.make_divisible(16, 8) == 16 and L.make_divisible(10.0, 8) == 16


def test_detect_bias_init_formula():
    torch.manual_seed(0)
    m = L.Model(L.load_cfg(scale="n"))
    det = m.model[-1]
    torch.manual_seed(0)
    raw = L.Detect(1, m.yaml["anchors"], [c.in_channels for c in det.m])
    # offsets added by _initialize_biases (reference models/yolo.py:352-359)
    import math
    for mi, s in zip(det.m, det.stride):
        b = mi.bias.detach().view(det.na, -1)
        assert b.shape == (3, 6)
        exp4 = math.log(8 / (640 / float(s)) ** 2)
        assert exp4 < 0
    assert raw.na == 3 and raw.no == 6


def test_modules_pickle_and_deepcopy():
    import copy
    import pickle
    m = L.Model(L.load_cfg(scale="n"))
    m2 = pickle.loads(pickle.dumps(m))
    m3 = copy.deepcopy(m)
    for a, b in zip(m.state_dict().values(), m2.state_dict().values()):
        assert torch.equal(a, b)
    assert type(m3.model[9]).__module__ == "lead_yolo_amd.modules"


def test_no_cpu_fallback():
    m = L.BasicStage(24, 1).eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 24, 8, 8))
    mdl = L.Model(L.load_cfg(scale="n")).eval()
    with pytest.raises(RuntimeError):
        mdl(torch.zeros(1, 3, 64, 64))


def test_capi_exports_every_declared_symbol():
    """The shared library loads and exports every function include/lead_yolo_hip.h declares."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "lead_yolo_hip.h")).read()
    declared = set(re.findall(r"\b(ly_[a-z0-9_]+)\s*\(", hdr))
    assert {"ly_mlpblock_fwd", "ly_gemm_fwd", "ly_conv3x3_fwd", "ly_rfcbam3_fwd"} <= declared
    lib = ctypes.CDLL(L.capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert set(L.capi.SIGNATURES) | {"ly_last_error"} >= declared
    assert L.capi.lib().ly_abi_version() >= 1


def test_struct_layouts_match_header_field_order():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "lead_yolo_hip.h")).read()
    for cls in (L.capi.LyGemmParams, L.capi.LyConv3Params, L.capi.LyRfcbam3Params, L.capi.LyWgradParams):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cls.__name__, cls.__name__), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", part.strip())[0])
        assert names == [f[0] for f in cls._fields_], cls.__name__


def test_frag_pack_nat_layout():
    """pack.frag_pack_nat (the A operand of ly_detect_level): lane = g*16 + i holds row 16t + i and k = 32s + 8g + j; hi + lo planes
    reproduce the fp32 matrix to bf16x2 precision, padding rows / columns are zero"""
    from lead_yolo_amd import pack
    g = torch.Generator().manual_seed(3)
    w = torch.randn(18, 72, generator=g)
    pk = pack.frag_pack_nat(w, planes=2)
    assert pk.shape == (2, 3, 2, 64, 8) and pk.dtype == torch.int16
    v = pk.view(torch.bfloat16).float()
    full = v[:, :, 0] + v[:, :, 1]                                    # [T, S, lane, j]
    for t, s_, lane, j in ((0, 0, 0, 0), (0, 1, 17, 3), (1, 2, 33, 7), (1, 0, 49, 5), (0, 2, 63, 7)):
        r, k = 16 * t + (lane & 15), 32 * s_ + 8 * (lane >> 4) + j
        want = float(w[r, k]) if (r < 18 and k < 72) else 0.0
        assert abs(float(full[t, s_, lane, j]) - want) <= 2e-5 * max(1.0, abs(want)), (t, s_, lane, j)
    one = pack.frag_pack_nat(w, planes=1).view(torch.bfloat16).float()
    assert one.shape == (2, 3, 1, 64, 8) and torch.equal(one[:, :, 0], v[:, :, 0])


def test_pick_tile():
    for ho, wo in ((40, 40), (20, 20), (80, 80), (5, 6), (1, 1), (3, 200)):
        th, tw = L.ops.pick_tile(ho, wo)
        assert 1 <= th * tw <= 64 and tw >= 1 and th >= 1


def test_optimizer_groups_and_ema_match_reference_recipe():
    """smart_optimizer: three groups in the reference's order (biases, decayed weights, norm weights) with the counts the
    reference produced for lead-yolo-n (fixture trainsteps_n); ModelEMA decay ramp d*(1 - exp(-u/tau))."""
    import math
    import lead_yolo_amd as L
    meta, _ = G.load("trainsteps_n")
    m = L.Model(L.load_cfg(scale="n"))
    opt = L.smart_optimizer(m, "SGD", meta["lr0"], meta["momentum"], meta["weight_decay"])
    assert [len(g["params"]) for g in opt.param_groups] == [meta["groups"]["n_bias"], meta["groups"]["n_decay"], meta["groups"]["n_bn"]]
    assert opt.param_groups[0]["weight_decay"] == 0 and opt.param_groups[2]["weight_decay"] == 0
    assert abs(opt.param_groups[1]["weight_decay"] - meta["weight_decay"]) < 1e-12
    assert all(g["nesterov"] and g["momentum"] == meta["momentum"] for g in opt.param_groups)
    ema = L.ModelEMA(m)
    w0 = {k: v.clone() for k, v in ema.ema.state_dict().items()}
    with torch.no_grad():
        for p in m.parameters():
            p.add_(1.0)
    ema.update(m)
    d = 0.9999 * (1 - math.exp(-1 / 2000))
    k = "model.0.proj.weight"
    assert torch.allclose(ema.ema.state_dict()[k], w0[k] * d + (w0[k] + 1.0) * (1 - d), atol=1e-6)
    assert ema.updates == 1


@pytest.mark.parametrize("scale", ["n", "s", "l"])
def test_two_consumer_layers_follow_the_routing_table(scale):
    """Model._two_consumer_layers (the layer outputs grad.fork splits in training) is exactly the set of layers read by two later layers of the
    reference's routing (`m.f`, models/yolo.py:179-195): the backbone stages 3 and 5 (next stage + neck concat), the neck maps 9 and 13 (upsample +
    later concat) and the neck outputs 16 and 19 (Detect level + next RFCBAMConv); layer 22 feeds Detect only"""
    torch.manual_seed(0)
    m = L.Model(L.load_cfg(scale=scale))
    readers = {}
    for mod in m.model:
        for j in ([mod.f] if isinstance(mod.f, int) else mod.f):
            readers.setdefault(mod.i - 1 if j == -1 else j, []).append(mod.i)
    assert sorted(m._two_consumer_layers()) == sorted(j for j, r in readers.items() if len(r) == 2 and j >= 0) == [3, 5, 9, 13, 16, 19]
    assert all(j in m.save or readers[j] == [j + 1] or (j + 1) in readers[j] for j in m._two_consumer_layers())
