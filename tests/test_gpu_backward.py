"""GPU parity tests of the training-step backward: gradients of the HIP modules vs the
reference's own vectors (dx_train, per-parameter gradient norms in the fixtures) and vs autograd through the
oracle on identical inputs.  Tolerance: 1e-3 relative to the largest magnitude of each gradient tensor."""
import copy

import numpy as np
import pytest
import torch

from oracle import synth
from tests import golden_util as G
from tests.test_gpu_modules import _bn_eps, _ctor, _dev, _load
from tests.test_oracle_golden import _run

pytestmark = pytest.mark.gpu

RTOL = 1e-3


def _close(got, want, what, rtol=RTOL, floor=0.0):
    got = got.detach().float().cpu().numpy()
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = float(np.abs(want).max())
    err = float(np.abs(got - want).max())
    assert err <= rtol * scale + floor + 1e-6, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def _floor(grads):
    """absolute slack for gradients that are zero in exact arithmetic (a bias in front of a train-mode BatchNorm):
    rounding noise there scales with the other gradients of the module, not with the (zero) value itself"""
    return 1e-5 * max(float(v.abs().max()) for v in grads.values())


def _oracle_grads(kind, ctor, st, x, r):
    st = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in st.items()}
    xt = x.clone().requires_grad_(True)
    y = _run(kind, list(ctor), st, xt, True)[0]
    (y * r).sum().backward()
    return y.detach(), xt.grad, {k: v.grad for k, v in st.items() if v.is_floating_point() and v.requires_grad}


def _hip_grads(m, x, r):
    xt = x.to(_dev()).requires_grad_(True)
    y = m(xt)
    (y * r.to(_dev())).sum().backward()
    return y.detach(), xt.grad, {k: p.grad for k, p in m.named_parameters()}


BACKWARD_MODULES = [n for n in (G.names("basicstage") + G.names("patchmerge") + G.names("coordatt") + G.names("cabottleneck") + G.names("c3ca")
                                + G.names("sppf") + G.names("rfcbam"))]


@pytest.mark.parametrize("name", BACKWARD_MODULES)
def test_module_backward_golden(name):
    """dx and every parameter gradient of a train-mode step, vs the reference (fixture) and the oracle (elementwise)"""
    meta, arr = G.load(name)
    st = G.state_for(meta)
    x = synth.synth_input(meta["in_shape"], meta["seed"] + 1)
    m = _bn_eps(_load(_ctor(meta["kind"])(*meta["ctor"]), st)).to(_dev()).train()
    r = synth.synth_input(arr["y_train"].shape, meta["seed"] + 2)
    y, dx, gp = _hip_grads(m, x, r)
    _close(y, arr["y_train"], name + " y_train")
    _close(dx, arr["dx_train"], name + " dx")
    nfloor = 1e-4 * max(meta["grad_norms"].values())      # slack for gradients that are exactly zero in exact arithmetic
    for k, want in meta["grad_norms"].items():
        assert gp[k] is not None, f"{name}: no gradient for {k}"
        got = float(gp[k].double().norm())
        assert abs(got - want) <= 2e-3 * want + nfloor + 1e-5, f"{name} |d{k}| = {got:.6e}, reference {want:.6e}"
    _, dxo, gpo = _oracle_grads(meta["kind"], meta["ctor"], st, x, r)
    for k, want in gpo.items():
        _close(gp[k], want, f"{name} d{k}", floor=_floor(gpo))
    sd = m.state_dict()
    for k in arr:
        if k.startswith("post_"):
            _close(sd[k[5:]], arr[k], name + " " + k)


def test_patchembed_backward_golden():
    """PatchEmbed reads the NCHW image: parameter gradients only (the image needs no gradient)"""
    name = "patchembed_3_24"
    meta, arr = G.load(name)
    st = G.state_for(meta)
    x = synth.synth_input(meta["in_shape"], meta["seed"] + 1)
    m = _bn_eps(_load(_ctor(meta["kind"])(*meta["ctor"]), st)).to(_dev()).train()
    r = synth.synth_input(arr["y_train"].shape, meta["seed"] + 2)
    y = m(x.to(_dev()))
    (y * r.to(_dev())).sum().backward()
    _close(y, arr["y_train"], name + " y_train")
    _, _, gpo = _oracle_grads(meta["kind"], meta["ctor"], st, x, r)
    for k, p in m.named_parameters():
        _close(p.grad, gpo[k], f"{name} d{k}", floor=_floor(gpo))


BWD_CASES = [
    ("BasicStage", (24, 1), (2, 24, 40, 36)),
    ("BasicStage", (40, 1), (3, 40, 17, 13)),
    ("BasicStage", (160, 1), (2, 160, 20, 20)),
    ("PatchMerging_FasterNet", (40, 80, 2, 2), (2, 40, 24, 20)),
    ("C3_CA", (168, 128, 1, False), (1, 168, 40, 40)),
    ("C3_CA", (64, 64, 3, True), (2, 64, 13, 11)),
    ("SPPF", (160, 160, 5), (2, 160, 20, 20)),
    ("RFCBAMConv", (160, 256, 1, 1), (2, 160, 20, 20)),
    ("RFCBAMConv", (128, 128, 3, 2), (2, 128, 40, 40)),
    ("RFCBAMConv", (64, 64, 3, 2), (1, 64, 21, 13)),
]


@pytest.mark.parametrize("kind,ctor,shape", BWD_CASES)
def test_module_backward_shapes_vs_oracle(kind, ctor, shape):
    """real layer shapes and ragged sizes: every gradient elementwise against autograd through the oracle"""
    torch.manual_seed(0)
    m = _ctor(kind)(*ctor)
    st = synth.synth_state(synth.shapes_of(m.state_dict()), 9100 + sum(shape) + len(kind))
    _bn_eps(_load(m, st))
    x = synth.synth_input(shape, 37 + shape[1])
    with torch.no_grad():
        y0 = _run(kind, list(ctor), copy.deepcopy(st), x.clone(), True)[0]
    r = synth.synth_input(tuple(y0.shape), 41 + shape[1])
    # no cotangent on outputs within 1e-4 of a ReLU kink: whether such an element is "on" is decided by rounding
    # (5e-6 forward differences), and one flipped element changes every gradient by O(1) * r there
    r = r * (y0.abs() > 1e-4)
    yo, dxo, gpo = _oracle_grads(kind, ctor, st, x, r)
    y, dx, gp = _hip_grads(m.to(_dev()).train(), x, r)
    _close(y, yo, f"{kind}{ctor} y")
    _close(dx, dxo, f"{kind}{ctor} dx")
    for k, want in gpo.items():
        _close(gp[k], want, f"{kind}{ctor} d{k}", floor=_floor(gpo))
