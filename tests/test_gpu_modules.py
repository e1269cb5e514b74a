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
                                     (16, 1, 5, 7), (80, 1, 3, 200), (160, 2, 1, 1), (40, 1, 64, 2), (320, 1, 10, 12)])
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
