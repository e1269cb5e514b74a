"""The eval pipeline of val.py:212-234 / detect.py:120-149 end to end on real dataset images — uint8 letterboxed batch -> /255 -> model
(eval) -> non_max_suppression -> boxes — HIP path vs the CPU oracle on the same seeded random-init weights, on 16 letterboxed images of the
SSDD test split that ships with the reference (tests/golden/ssdd16.npz, oracle/gen_ssdd_fixture.py; data/SSDD.yaml is the nc = 1 dataset
the LEAD-YOLO recipe trains on), plus the training loss on the dataset's own labels (real boxes: small, clustered ships, ~2 per image).
The NMS step of the oracle (oracle/nms.py) restates torchvision.ops.nms' documented contract — torchvision is not in the image, so that one
step stays parity-unpinned; everything in front of it is pinned by the reference's own vectors."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import functional as OF
from oracle import metrics as OMET
from oracle import nms as ONMS
from oracle import synth
from tests.test_gpu_modules import _cfg, _dev

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ssdd16.npz")


def _batch():
    d = np.load(FIX)
    g = torch.from_numpy(d["imgs"])                              # [16, 320, 320] uint8 luma, letterboxed (pad value 114)
    return g.unsqueeze(1).expand(-1, 3, -1, -1).contiguous(), torch.from_numpy(d["targets"])


def test_fixture_is_a_letterboxed_dataset_batch():
    imgs, tg = _batch()
    assert imgs.dtype == torch.uint8 and tuple(imgs.shape) == (16, 3, 320, 320)
    assert int((imgs[:, 0, 0, :] == 114).all(1).sum()) >= 12             # landscape chips: the top rows are letterbox padding
    assert tg.shape[1] == 6 and tg.shape[0] >= 16 and float(tg[:, 2:].min()) > 0 and float(tg[:, 2:].max()) < 1
    assert set(tg[:, 0].long().tolist()) <= set(range(16)) and (tg[:, 1] == 0).all()


def _iou(a, b):
    lt = np.maximum(a[:, None, :2], b[None, :, :2]); rb = np.minimum(a[:, None, 2:4], b[None, :, 2:4])
    inter = np.clip(rb - lt, 0, None).prod(2)
    area = lambda x: (x[:, 2] - x[:, 0]) * (x[:, 3] - x[:, 1])
    return inter / (area(a)[:, None] + area(b)[None, :] - inter + 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("conf", [0.001, 0.02])
def test_val_pipeline_boxes_match_oracle(conf):
    """conf 0.001 / iou 0.6 are val.py's settings (thousands of candidates per image), 0.02 a sparser set.  Boxes are compared as sets:
    the two forwards differ by ~1e-5 (bf16x3 products), so candidates whose scores nearly tie may swap order and NMS may then keep the
    neighbour — at least 97 % of the oracle's boxes must have a partner with IoU > 0.98 and |conf| within 1e-3, and the counts must agree
    to 3 %."""
    import lead_yolo_amd as L
    imgs, _ = _batch()
    torch.manual_seed(0)
    cfg = _cfg("s")
    m = L.Model(cfg)
    st = synth.synth_state(synth.shapes_of(m.state_dict()), 6262)
    st["model.23.anchors"] = m.model[-1].anchors.clone()
    st["model.23.m.0.bias"] = st["model.23.m.0.bias"] + 2.0              # (random heads score ~0.5 * 0.5: lift a few over the thresholds)
    m.load_state_dict(st)
    x = imgs.float() / 255                                                # val.py:212-214
    with torch.no_grad():
        zo, _ = OF.model_forward(copy.deepcopy(st), cfg, x, m.stride, training=False)
        z, _ = m.to(_dev()).eval()(x.to(_dev()))
    want, _ = ONMS.non_max_suppression(zo.numpy(), conf, 0.6)
    got = L.non_max_suppression(z, conf, 0.6)                             # val.py:230-234
    total = matched = 0
    for i in range(16):
        g, w = got[i].cpu().numpy(), want[i]
        assert abs(len(g) - len(w)) <= max(2, 0.03 * len(w)), (i, len(g), len(w))
        if len(w) == 0:
            continue
        iou = _iou(w, g) if len(g) else np.zeros((len(w), 0))
        for r in range(len(w)):
            ok = (iou[r] > 0.98) & (np.abs(g[:, 4] - w[r, 4]) < 1e-3)
            matched += bool(ok.any())
        total += len(w)
    assert total > 200 and matched >= 0.97 * total, (matched, total)


@pytest.mark.gpu
def test_loss_on_dataset_labels():
    """ComputeLoss on the dataset's own labels (train.py:317-318): device loss on the HIP model's raw maps vs the oracle loss on the oracle's"""
    import lead_yolo_amd as L
    imgs, tg = _batch()
    torch.manual_seed(0)
    cfg = _cfg("n")
    m = L.Model(cfg)
    st = synth.synth_state(synth.shapes_of(m.state_dict()), 6363)
    st["model.23.anchors"] = m.model[-1].anchors.clone()
    m.load_state_dict(st)
    x = imgs.float() / 255
    with torch.no_grad():
        _, po = OF.model_forward(copy.deepcopy(st), cfg, x, m.stride, training=False)
        lo, io = OF.compute_loss(po, tg, st["model.23.anchors"], nc=1)
        _, p = m.to(_dev()).eval()(x.to(_dev()))
        l, it = L.ComputeLoss(m)(p, tg.to(_dev()))
    assert abs(float(l) - float(lo)) <= 1e-3 * abs(float(lo)), (float(l), float(lo))
    np.testing.assert_allclose(it.cpu().numpy(), io.numpy(), rtol=1e-3, atol=1e-6)


STEPS = 600


def _stats_of(boxes_per_image, tg, size):
    """val.py:127-166 for letterboxed images scored at the letterboxed size (no rescale to native space): per image
    (correct [n, 10], conf, class, target classes)"""
    out = []
    for i, pred in enumerate(boxes_per_image):
        lab = tg[tg[:, 0] == i]
        tcls = lab[:, 1].numpy()
        pred = np.asarray(pred, np.float32).reshape(-1, 6)
        if len(pred) == 0:
            if len(lab):
                out.append((np.zeros((0, 10), bool), np.zeros(0, np.float32), np.zeros(0, np.float32), tcls))
            continue
        correct = np.zeros((len(pred), 10), bool)
        if len(lab):
            tbox = ONMS.xywh2xyxy(lab[:, 2:6].numpy().astype(np.float32)) * size             # val.py:160 (labels are normalised xywh)
            correct = OMET.process_batch(pred, np.concatenate((lab[:, 1:2].numpy().astype(np.float32), tbox), 1))
        out.append((correct, pred[:, 4], pred[:, 5], tcls))
    return out


TRAIN_FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ssdd_train48.npz")


def _train_batches():
    """the 48 TRAIN-split images (tests/golden/ssdd_train48.npz, oracle/gen_ssdd_fixture.py) as three batches of 16 with batch-local image indices"""
    d = np.load(TRAIN_FIX)
    g = torch.from_numpy(d["imgs"]).unsqueeze(1).expand(-1, 3, -1, -1).contiguous()
    tg = torch.from_numpy(d["targets"])
    out = []
    for b in range(3):
        rows = tg[(tg[:, 0] >= 16 * b) & (tg[:, 0] < 16 * b + 16)].clone()
        rows[:, 0] -= 16 * b
        out.append((g[16 * b:16 * b + 16].contiguous(), rows))
    return out


def test_train_and_test_fixtures_are_disjoint_splits():
    a, b = np.load(TRAIN_FIX), np.load(FIX)
    assert a["imgs"].shape == (48, 320, 320) and not (set(a["names"].tolist()) & set(b["names"].tolist()))


@pytest.mark.gpu
def test_ssdd_short_training_then_heldout_map_hip_vs_oracle():
    """Accuracy evidence at the metric level on the dataset the recipe trains on (data/SSDD.yaml: 928 train / 232 test images, val.py:183-188),
    HELD OUT: lead-yolo-n is trained for STEPS optimisation steps of the HIP training step (train.py:295-341: uint8 batch, ComputeLoss, clip,
    SGD-nesterov with the reference's warm-up, EMA) on 48 letterboxed images of the TRAIN split, then mAP@0.5 / mAP@0.5:0.95 of the trained
    EMA weights is computed on the 16 images of the TEST split (none of them seen in training) twice — HIP eval forward + device NMS, and
    fp32 CPU oracle forward + oracle NMS — with the reference's metric code (oracle/metrics.py, pinned by tests/golden/metrics_cases.npz).
    The two pipelines must agree to 0.02 mAP, the loss must have fallen, and the detector must find ships it has never seen (held-out
    mAP@0.5 far above the ~0 of the initial weights; the figure on its own training images is printed beside it)."""
    import lead_yolo_amd as L
    from lead_yolo_amd import train as T
    imgs, tg = _batch()                                                      # the held-out test images
    batches = [(x.to(_dev()), t.to(_dev())) for x, t in _train_batches()]
    torch.manual_seed(0)
    cfg = _cfg("n")
    m = L.Model(cfg).to(_dev()).train()                                     # the model's own initialisation (models/yolo.py:213-233)
    loss_fn = L.ComputeLoss(m)
    opt = T.smart_optimizer(m, "SGD", 0.01, 0.937, 5e-4)
    ema = T.ModelEMA(m)
    losses = []
    for it in range(STEPS):
        lr = 0.01 * min(1.0, (it + 1) / 30)                                   # linear warm-up of train.py:300-307, then the base rate
        for gi, g in enumerate(opt.param_groups):
            g["lr"] = max(0.1 - 0.09 * it / 30, lr) if gi == 0 else lr        # the bias group (first, utils/torch_utils.py:337) warms DOWN from 0.1
        xb, tb = batches[it % 3]
        loss, _ = T.train_step(m, loss_fn, opt, xb, tb, ema=ema)
        losses.append(float(loss))
    assert np.isfinite(losses).all() and np.mean(losses[-10:]) < 0.6 * np.mean(losses[:10]), (losses[:3], losses[-3:])
    me = ema.ema.eval()
    sd = {k: v.detach().float().cpu() for k, v in me.state_dict().items()}

    def score(images, targets, oracle):
        x = images.float() / 255
        with torch.no_grad():
            if oracle:
                zo, _ = OF.model_forward(copy.deepcopy(sd), cfg, x, m.stride.cpu(), training=False)
                boxes, _ = ONMS.non_max_suppression(zo.numpy(), 0.001, 0.6)
            else:
                z, _ = me(x.to(_dev()))
                boxes = [b.cpu().numpy() for b in L.non_max_suppression(z, 0.001, 0.6)]      # val.py:230-234 settings
        return OMET.mean_results(_stats_of(boxes, targets, 320))

    rg, rw = score(imgs, tg, False), score(imgs, tg, True)
    seen = score(*_train_batches()[0], False)
    print(f"SSDD after {STEPS} steps on 48 train images: loss {np.mean(losses[:10]):.3f} -> {np.mean(losses[-10:]):.3f}; held-out (P, R, mAP50, mAP) "
          f"HIP {rg} oracle {rw}; on 16 of its training images HIP {seen}")
    assert abs(rg[2] - rw[2]) <= 0.02 and abs(rg[3] - rw[3]) <= 0.02, (rg, rw)
    assert rg[2] > HELDOUT_MAP50_MIN and seen[2] > 0.9, ("the trained detector does not generalise / has not learnt its training images", rg, seen, losses[-3:])


HELDOUT_MAP50_MIN = 0.25          # measured on MI355X (round 6): see the print above; the initial weights score ~0
