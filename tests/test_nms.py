"""Eval tail: non_max_suppression.  CPU: the oracle restatement (oracle/nms.py) against hand-computed cases and the defining properties of
greedy NMS.  GPU: the HIP path (lead-yolo_amd/nms.py -> csrc/ly_nms.hip) against the oracle, kept indices bit-exact."""
import numpy as np
import pytest
import torch

from oracle import nms as ON


def _iou(a, b):
    iw = max(min(a[2], b[2]) - max(a[0], b[0]), 0.0)
    ih = max(min(a[3], b[3]) - max(a[1], b[1]), 0.0)
    inter = iw * ih
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def _random_pred(bs, n, nc, seed, cluster=True):
    g = np.random.default_rng(seed)
    p = np.zeros((bs, n, 5 + nc), np.float32)
    centres = g.uniform(50, 590, (bs, 12, 2))
    which = g.integers(0, 12, (bs, n))
    xy = np.take_along_axis(centres, which[..., None].repeat(2, -1), 1) + g.normal(0, 6 if cluster else 200, (bs, n, 2))
    p[..., 0:2] = xy
    p[..., 2:4] = g.uniform(20, 120, (bs, n, 2))
    p[..., 4] = g.uniform(0, 1, (bs, n)) ** 2
    p[..., 5:] = g.uniform(0, 1, (bs, n, nc))
    return p


def test_oracle_hand_case():
    # three boxes of one class: the second overlaps the first (IoU 0.68 > 0.45) and is dropped, the third is apart
    pred = np.array([[[50, 50, 20, 20, 0.9, 1.0], [52, 52, 20, 20, 0.8, 1.0], [100, 100, 20, 20, 0.7, 1.0], [10, 10, 5, 5, 0.1, 1.0]]], np.float32)
    out, kept = ON.non_max_suppression(pred, 0.25, 0.45)
    assert kept[0].tolist() == [0, 2]
    np.testing.assert_allclose(out[0], [[40, 40, 60, 60, 0.9, 0], [90, 90, 110, 110, 0.7, 0]], rtol=0, atol=1e-6)
    # different classes do not suppress each other (class offset), unless agnostic
    pred2 = np.array([[[50, 50, 20, 20, 0.9, 1.0, 0.0], [52, 52, 20, 20, 0.8, 0.0, 1.0]]], np.float32)
    assert ON.non_max_suppression(pred2, 0.25, 0.45)[1][0].tolist() == [0, 1]
    assert ON.non_max_suppression(pred2, 0.25, 0.45, agnostic=True)[1][0].tolist() == [0]
    assert ON.non_max_suppression(pred2, 0.25, 0.45, classes=[1])[1][0].tolist() == [1]
    assert ON.non_max_suppression(pred2, 0.25, 0.45, max_det=1)[1][0].tolist() == [0]


@pytest.mark.parametrize("nc,agnostic", [(1, False), (3, False), (3, True)])
def test_oracle_greedy_properties(nc, agnostic):
    pred = _random_pred(2, 600, nc, 7)
    conf, thr = 0.2, 0.45
    out, kept = ON.non_max_suppression(pred, conf, thr, agnostic=agnostic, max_det=10000)
    for b in range(2):
        d = out[b]
        assert d.shape[0] > 5
        assert np.all(np.diff(d[:, 4]) <= 0)                                   # sorted by confidence
        off = 0.0 if agnostic else ON.MAX_WH
        bx = d[:, :4] + d[:, 5:6] * off
        for i in range(len(d)):                                               # kept boxes do not suppress each other
            for j in range(i):
                assert _iou(bx[j], bx[i]) <= thr + 1e-6
        # every candidate that was dropped overlaps a kept box of higher (or equal) confidence
        x = pred[b]
        cand = np.nonzero(x[:, 4] > conf)[0]
        sc = (x[cand, 5:] * x[cand, 4:5])
        cls = sc.argmax(1)
        sc = sc.max(1)
        ok = sc > conf
        cand, sc, cls = cand[ok], sc[ok], cls[ok]
        boxes = ON.xywh2xyxy(x[cand, :4]) + cls[:, None].astype(np.float32) * off
        kept_set = set(kept[b].tolist())
        for i, ci in enumerate(cand):
            if ci in kept_set:
                continue
            assert any(_iou(bk, boxes[i]) > thr - 1e-6 and sk >= sc[i] for bk, sk in zip(bx, d[:, 4]))


@pytest.mark.gpu
@pytest.mark.parametrize("nc,agnostic,classes,conf,max_det", [(1, False, None, 0.25, 300), (1, False, None, 0.001, 300), (3, False, None, 0.2, 300),
                                                                (3, True, None, 0.2, 50), (4, False, [0, 2], 0.1, 300)])
def test_hip_nms_matches_oracle(nc, agnostic, classes, conf, max_det):
    import lead_yolo_amd as L
    dev = torch.device("cuda:0")
    pred = _random_pred(3, 2500, nc, 11 + nc)
    want, want_idx = ON.non_max_suppression(pred, conf, 0.45, classes=classes, agnostic=agnostic, max_det=max_det)
    got = L.non_max_suppression(torch.from_numpy(pred).to(dev), conf, 0.45, classes=classes, agnostic=agnostic, max_det=max_det)
    dets, count, keep = L.nms_padded(torch.from_numpy(pred).to(dev), conf, 0.45, classes=classes, agnostic=agnostic, max_det=max_det)
    for b in range(3):
        assert int(count[b]) == len(want_idx[b])
        assert keep[b, :len(want_idx[b])].cpu().tolist() == want_idx[b].tolist()           # the same boxes, in the same order: bit-exact
        np.testing.assert_array_equal(got[b].cpu().numpy(), want[b])
        assert float(dets[b, len(want_idx[b]):].abs().sum()) == 0.0


def test_oracle_multi_label_hand_case():
    """multi_label (utils/general.py:921, 951-955): a box whose two class scores both pass becomes two detections (one per class, class-offset
    boxes do not suppress each other); with best-class-only it is one"""
    pred = np.array([[[50, 50, 20, 20, 1.0, 0.9, 0.8, 0.05],            # classes 0 and 1 pass
                      [52, 50, 20, 20, 1.0, 0.1, 0.7, 0.05],            # class 1 only: overlaps the first box, lower score -> suppressed within class 1
                      [150, 150, 10, 10, 0.9, 0.05, 0.05, 0.6]]], np.float32)
    out, idx = ON.non_max_suppression(pred, 0.25, 0.45, multi_label=True)
    assert idx[0].tolist() == [0 * 3 + 0, 0 * 3 + 1, 2 * 3 + 2]
    np.testing.assert_allclose(out[0][:, 4], [0.9, 0.8, 0.54], rtol=1e-6)
    assert out[0][:, 5].tolist() == [0.0, 1.0, 2.0]
    out1, idx1 = ON.non_max_suppression(pred, 0.25, 0.45, multi_label=False)
    assert idx1[0].tolist() == [0, 1, 2] and out1[0][:, 5].tolist() == [0.0, 1.0, 2.0]      # best class only: box 1 (class 1) survives, no class-1 twin of box 0


@pytest.mark.gpu
@pytest.mark.parametrize("nc,agnostic,classes,conf,max_det", [(3, False, None, 0.2, 300), (5, True, None, 0.15, 100), (4, False, [1, 3], 0.1, 300),
                                                                (2, False, None, 0.001, 300)])
def test_hip_multi_label_nms_matches_oracle(nc, agnostic, classes, conf, max_det):
    """val.py's NMS setting for nc > 1 (multi_label=True): kept (box, class) pairs bit-exact vs the oracle, in the same order"""
    import lead_yolo_amd as L
    dev = torch.device("cuda:0")
    pred = _random_pred(3, 2000, nc, 31 + nc)
    want, want_idx = ON.non_max_suppression(pred, conf, 0.45, classes=classes, agnostic=agnostic, max_det=max_det, multi_label=True)
    got = L.non_max_suppression(torch.from_numpy(pred).to(dev), conf, 0.45, classes=classes, agnostic=agnostic, max_det=max_det, multi_label=True)
    _, count, keep = L.nms_padded(torch.from_numpy(pred).to(dev), conf, 0.45, classes=classes, agnostic=agnostic, max_det=max_det, multi_label=True)
    total = 0
    for b in range(3):
        assert int(count[b]) == len(want_idx[b])
        assert keep[b, :len(want_idx[b])].cpu().tolist() == want_idx[b].tolist()
        np.testing.assert_array_equal(got[b].cpu().numpy(), want[b])
        total += len(want_idx[b])
    assert total > 20


@pytest.mark.gpu
def test_hip_nms_edge_cases():
    import lead_yolo_amd as L
    dev = torch.device("cuda:0")
    empty = torch.zeros((2, 100, 6), device=dev)
    out = L.non_max_suppression(empty)
    assert [tuple(o.shape) for o in out] == [(0, 6), (0, 6)]
    one = torch.tensor([[[50.0, 50, 20, 20, 0.9, 1.0]]], device=dev)
    out = L.non_max_suppression(one)
    np.testing.assert_allclose(out[0].cpu().numpy(), [[40, 40, 60, 60, 0.9, 0]], atol=1e-6)
    assert [tuple(o.shape) for o in L.non_max_suppression(torch.zeros((1, 10, 8), device=dev), multi_label=True)] == [(0, 6)]
    with pytest.raises(NotImplementedError):
        L.non_max_suppression(torch.zeros((1, 10, 8), device=dev), labels=[torch.zeros(1, 5)])
    with pytest.raises(RuntimeError):
        L.non_max_suppression(torch.zeros((1, 10, 6)))
    # model in validation mode hands (inference_out, loss_out)
    out = L.non_max_suppression((one, None))
    assert out[0].shape == (1, 6)
