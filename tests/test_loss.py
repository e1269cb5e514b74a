"""Product loss / target assignment (lead_yolo_amd.loss) vs the vectors the reference produced:
int64 indices bit-exact, loss and input gradients to fp32 rounding.  On the CPU the torch formulation runs; on the GPU
`ComputeLoss.__call__` is the fused device loss (csrc/ly_loss.hip: forward and gradient in 3 launches per level), while
`build_targets` (the inspectable index API) is the same host logic on both."""
import numpy as np
import pytest
import torch

from tests import golden_util as G


class _Det:
    def __init__(self, anchors):
        self.na, self.nc, self.nl, self.anchors = anchors.shape[1], 1, anchors.shape[0], anchors


def _run(case, device):
    from lead_yolo_amd.loss import ComputeLoss
    meta, arr = G.load("loss_n")
    anchors = G.t(arr["anchors"]).to(device)
    cl = ComputeLoss(_Det(anchors), hyp={k: v for k, v in meta["hyp"].items() if k in ("box", "cls", "cls_pw", "obj", "obj_pw", "anchor_t", "fl_gamma")})
    preds = [G.t(arr[f"pred{i}"]).to(device).requires_grad_(True) for i in range(3)]
    tg = G.t(arr[f"{case}_targets"]).to(device)
    tcls, tbox, indices, anch = cl.build_targets(preds, tg)
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
        # where the last assignment wins; the fused device loss (csrc/ly_loss.hip) elects the same winner (highest candidate
        # index), so the GPU result matches cell for cell — unlike torch's unordered scatter on a GPU.
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("case", ["rand", "edge", "empty"])
def test_loss_cpu(case):
    _run(case, torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["rand", "edge", "empty"])
def test_loss_gpu(case):
    _run(case, torch.device("cuda:0"))


@pytest.mark.gpu
@pytest.mark.parametrize("bs,nt,seed", [(8, 200, 0), (4, 1, 1), (16, 900, 2)])
def test_fused_loss_vs_torch_formulation(bs, nt, seed):
    """the device loss against the torch formulation run on the CPU (sequential scatter = the reference's semantics) on random
    predictions with many targets per cell: total, items and the gradient of every prediction"""
    from lead_yolo_amd.loss import ComputeLoss
    _, arr = G.load("loss_n")
    anchors = G.t(arr["anchors"])
    g = torch.Generator().manual_seed(seed)
    preds = [torch.randn(bs, 3, s, s, 6, generator=g) for s in (40, 20, 10)]
    tg = torch.cat((torch.randint(0, bs, (nt, 1), generator=g).float(), torch.zeros(nt, 1), torch.rand(nt, 2, generator=g),
                    torch.rand(nt, 2, generator=g) * 0.4 + 0.01), 1)
    res = []
    nthreads = torch.get_num_threads()
    for dev in (torch.device("cpu"), torch.device("cuda:0")):
        cl = ComputeLoss(_Det(anchors.to(dev)))
        ps = [p.clone().to(dev).requires_grad_(True) for p in preds]
        # torch's CPU index_put splits more than 3000 indices over threads; one thread keeps `tobj[...] = iou` sequential
        torch.set_num_threads(1)
        try:
            loss, items = cl(ps, tg.to(dev))
            loss.backward()
        finally:
            torch.set_num_threads(nthreads)
        res.append((loss.detach().cpu(), items.cpu(), [p.grad.cpu() for p in ps]))
    (l0, i0, g0), (l1, i1, g1) = res
    np.testing.assert_allclose(l1.numpy(), l0.numpy(), rtol=1e-4)
    np.testing.assert_allclose(i1.numpy(), i0.numpy(), rtol=1e-4, atol=1e-6)
    for a, b in zip(g1, g0):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=5e-4, atol=2e-6)
