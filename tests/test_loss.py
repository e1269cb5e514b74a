"""Device loss / target assignment (lead_yolo_amd.loss -> csrc/ly_loss.hip) vs the vectors the reference produced:
int64 indices bit-exact, loss and input gradients to fp32 rounding; and vs the oracle (oracle/functional.py, CPU) on random
many-targets-per-cell cases.  The product has no CPU loss: the CPU side of these tests is the oracle."""
import numpy as np
import pytest
import torch

from tests import golden_util as G


class _Det:
    def __init__(self, anchors):
        self.na, self.nc, self.nl, self.anchors = anchors.shape[1], 1, anchors.shape[0], anchors


def test_loss_refuses_cpu():
    from lead_yolo_amd.loss import ComputeLoss
    _, arr = G.load("loss_n")
    cl = ComputeLoss(_Det(G.t(arr["anchors"])))
    preds = [G.t(arr[f"pred{i}"]) for i in range(3)]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cl(preds, G.t(arr["rand_targets"]))
    with pytest.raises(NotImplementedError):
        ComputeLoss(_Det(G.t(arr["anchors"])), hyp=dict(fl_gamma=1.5))


_HYP_KEYS = ("box", "cls", "cls_pw", "obj", "obj_pw", "anchor_t", "fl_gamma", "label_smoothing")


def test_oracle_multiclass_loss_matches_reference():
    """oracle/functional.compute_loss with nc = 3, label smoothing 0.1, cls_pw 1.3, obj_pw 0.8 vs the reference's own vectors
    (tests/golden/loss_nc3.npz, oracle/gen_golden.py gen_loss_multiclass; utils/loss.py:168-173)"""
    from oracle import functional as OF
    meta, arr = G.load("loss_nc3")
    preds = [G.t(arr[f"pred{i}"]).requires_grad_(True) for i in range(3)]
    loss, items = OF.compute_loss(preds, G.t(arr["targets"]), G.t(arr["anchors"]), nc=3, hyp={k: meta["hyp"][k] for k in _HYP_KEYS})
    loss.backward()
    np.testing.assert_allclose(loss.detach().numpy(), arr["loss"], rtol=1e-5)
    np.testing.assert_allclose(items.numpy(), arr["items"], rtol=1e-5, atol=1e-7)
    assert arr["items"][2] > 0                       # the class term is really there
    for i in range(3):
        np.testing.assert_allclose(preds[i].grad.numpy(), arr[f"dpred{i}"], rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
def test_multiclass_loss_gpu():
    """the device loss with nc = 3 (class BCE, label smoothing, cls_pw, obj_pw — utils/loss.py:137-141, 168-173) vs the reference's
    vectors: loss, items, d/dpred of every level incl. the class logits, and build_targets' tcls bit-exact"""
    from lead_yolo_amd.loss import ComputeLoss
    device = torch.device("cuda:0")
    meta, arr = G.load("loss_nc3")
    det = _Det(G.t(arr["anchors"]).to(device))
    det.nc = 3
    cl = ComputeLoss(det, hyp={k: meta["hyp"][k] for k in _HYP_KEYS})
    preds = [G.t(arr[f"pred{i}"]).to(device).requires_grad_(True) for i in range(3)]
    tg = G.t(arr["targets"]).to(device)
    tcls, _, _, _ = cl.build_targets(preds, tg)
    for i in range(3):
        got = tcls[i].cpu().numpy()
        assert got.dtype == np.int64 and np.array_equal(got, arr[f"tcls{i}"])
    loss, items = cl(preds, tg)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), arr["loss"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(items.cpu().numpy(), arr["items"], rtol=2e-5, atol=1e-6)
    loss.backward()
    for i in range(3):
        np.testing.assert_allclose(preds[i].grad.cpu().numpy(), arr[f"dpred{i}"], rtol=2e-4, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["rand", "edge", "empty"])
def test_loss_gpu(case):
    from lead_yolo_amd.loss import ComputeLoss
    device = torch.device("cuda:0")
    meta, arr = G.load("loss_n")
    anchors = G.t(arr["anchors"]).to(device)
    cl = ComputeLoss(_Det(anchors), hyp={k: v for k, v in meta["hyp"].items() if k in _HYP_KEYS})
    preds = [G.t(arr[f"pred{i}"]).to(device).requires_grad_(True) for i in range(3)]
    tg = G.t(arr[f"{case}_targets"]).to(device)
    tcls, tbox, indices, anch = cl.build_targets(preds, tg)          # the matching kernel's own buffers, read back
    for i in range(3):
        got = np.stack([v.cpu().numpy() for v in indices[i]])
        assert got.dtype == np.int64 and np.array_equal(got, arr[f"{case}_idx{i}"])
        np.testing.assert_array_equal(tbox[i].detach().cpu().numpy(), arr[f"{case}_tbox{i}"])
        np.testing.assert_array_equal(anch[i].cpu().numpy(), arr[f"{case}_anch{i}"])
    loss, items = cl(preds, tg)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), arr[f"{case}_loss"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(items.cpu().numpy(), arr[f"{case}_items"], rtol=2e-5, atol=1e-6)
    loss.backward()
    for i in range(3):
        got, want = preds[i].grad.cpu().numpy(), arr[f"{case}_dpred{i}"]
        # `tobj[b, a, gj, gi] = iou` has duplicate cells (several targets per cell).  The reference's vectors come from the CPU,
        # where the last assignment wins; the device loss elects the same winner (highest candidate index), so the GPU result
        # matches cell for cell — unlike torch's unordered scatter on a GPU.
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("bs,nt,seed", [(8, 200, 0), (4, 1, 1), (16, 900, 2)])
def test_device_loss_vs_oracle(bs, nt, seed):
    """the device loss against the oracle on the CPU (sequential scatter = the reference's semantics) on random predictions
    with many targets per cell: total, items, the gradient of every prediction, and build_targets' indices bit-exact"""
    from lead_yolo_amd.loss import ComputeLoss
    from oracle import functional as OF
    _, arr = G.load("loss_n")
    anchors = G.t(arr["anchors"])
    g = torch.Generator().manual_seed(seed)
    preds = [torch.randn(bs, 3, s, s, 6, generator=g) for s in (40, 20, 10)]
    tg = torch.cat((torch.randint(0, bs, (nt, 1), generator=g).float(), torch.zeros(nt, 1), torch.rand(nt, 2, generator=g),
                    torch.rand(nt, 2, generator=g) * 0.4 + 0.01), 1)
    nthreads = torch.get_num_threads()
    ps = [p.clone().requires_grad_(True) for p in preds]
    torch.set_num_threads(1)          # torch's CPU index_put splits > 3000 indices over threads; one thread keeps `tobj[...] = iou` sequential
    try:
        l0, i0 = OF.compute_loss(ps, tg, anchors, nc=1)
        l0.backward()
    finally:
        torch.set_num_threads(nthreads)
    g0 = [p.grad for p in ps]
    dev = torch.device("cuda:0")
    cl = ComputeLoss(_Det(anchors.to(dev)))
    pd = [p.clone().to(dev).requires_grad_(True) for p in preds]
    l1, i1 = cl(pd, tg.to(dev))
    l1.backward()
    np.testing.assert_allclose(l1.detach().cpu().numpy(), l0.detach().numpy(), rtol=1e-4)
    np.testing.assert_allclose(i1.cpu().numpy(), i0.detach().numpy(), rtol=1e-4, atol=1e-6)
    for a, b in zip(pd, g0):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.numpy(), rtol=5e-4, atol=2e-6)
    _, tb1, idx1, an1 = cl.build_targets(pd, tg.to(dev))
    _, tb0, idx0, an0 = OF.build_targets([tuple(p.shape) for p in preds], tg, anchors)
    for i in range(3):
        for u, v in zip(idx1[i], idx0[i]):
            assert torch.equal(u.cpu(), v)
        assert torch.equal(tb1[i].cpu(), tb0[i]) and torch.equal(an1[i].cpu(), an0[i])


@pytest.mark.gpu
def test_out_of_range_image_index_is_loud_and_safe():
    """a target row whose image index is outside the batch (last partial batch, per-rank slices with global indices) must not
    index out of bounds: the row is rejected, the loss comes back NaN, build_targets raises IndexError (as torch indexing would)"""
    from lead_yolo_amd.loss import ComputeLoss
    _, arr = G.load("loss_n")
    dev = torch.device("cuda:0")
    cl = ComputeLoss(_Det(G.t(arr["anchors"]).to(dev)))
    preds = [G.t(arr[f"pred{i}"]).to(dev).requires_grad_(True) for i in range(3)]
    bs = preds[0].shape[0]
    tg = G.t(arr["rand_targets"]).to(dev).clone()
    guard = torch.full((1 << 20,), 7.0, device=dev)              # memory right after the allocations above keeps its contents
    for bad in (float(bs), float(bs + 1000), -2.0, float("nan")):
        t2 = tg.clone()
        t2[0, 0] = bad
        loss, _ = cl(preds, t2)
        assert torch.isnan(loss).all()
        with pytest.raises(IndexError):
            cl.build_targets(preds, t2)
    assert bool((guard == 7.0).all())
    loss, _ = cl(preds, tg)
    assert torch.isfinite(loss).all()
    # rows with image index exactly -1 are padding (fixed-shape target buffers of a captured training step): ignored
    pad = torch.cat((tg, torch.full((5, 6), -1.0, device=dev)))
    loss_p, items_p = cl(preds, pad)
    assert torch.equal(loss_p, loss)
