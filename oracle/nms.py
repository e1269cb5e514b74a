"""ORACLE — CPU restatement of the reference's eval tail `non_max_suppression` (TEST INFRASTRUCTURE, not product code).

Follows qingqing-zijin/LEAD-YOLO `utils/general.py:884-994` step by step for the options the detector's own callers use
(`detect.py:149`, `val.py:230-234`: single label per box, optional class filter, optional class-agnostic mode, `max_det`):
  candidates  obj > conf_thres                                             (:914)
  conf        x[:, 5:] *= x[:, 4:5]                                         (:944)
  box         xywh2xyxy                                                     (:947, utils/general.py:760-767)
  best class  conf, j = x[:, 5:mi].max(1); keep conf > conf_thres           (:955-956)
  class filter                                                              (:959-960)
  order       x[x[:, 4].argsort(descending=True)[:max_nms]]                 (:970)
  NMS         torchvision.ops.nms(boxes + cls * max_wh, scores, iou_thres)  (:973-975), then [:max_det] (:976)

PARITY UNPINNED for the last step: `torchvision` (a requirements.txt dependency of the reference, `torchvision>=0.8.1`) is NOT installed in
the build image, so the reference function itself cannot be executed here.  `nms` below restates torchvision's documented contract —
"iteratively removes lower scoring boxes which have an IoU greater than iou_threshold with another (higher scoring) box", boxes
(x1, y1, x2, y2), IoU = inter / (area_a + area_b - inter), result sorted by decreasing score — and the tests check the defining
properties (no two kept boxes of one class overlap above the threshold; every dropped candidate overlaps a kept, higher-scored one;
order; max_det) plus hand-computed cases.  Ties in score: torch's argsort does not define their order; this oracle (and the HIP path)
break them by ascending candidate index.

Only tests/ may import this module.
"""
import numpy as np

MAX_WH = 7680        # utils/general.py:918
MAX_NMS = 30000      # utils/general.py:919


def xywh2xyxy(x):
    """utils/general.py:760-767"""
    y = np.empty_like(x)
    y[..., 0] = x[..., 0] - x[..., 2] / 2
    y[..., 1] = x[..., 1] - x[..., 3] / 2
    y[..., 2] = x[..., 0] + x[..., 2] / 2
    y[..., 3] = x[..., 1] + x[..., 3] / 2
    return y


def nms(boxes, scores, iou_thres):
    """torchvision.ops.nms contract (greedy, IoU > threshold suppresses); `scores` must already be sorted descending.
    float32 arithmetic like the library's kernels.  Returns kept indices in order."""
    boxes = boxes.astype(np.float32)
    n = boxes.shape[0]
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    alive = np.ones(n, dtype=bool)
    keep = []
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(i)
        j = np.arange(i + 1, n)
        j = j[alive[i + 1:]]
        if j.size == 0:
            continue
        xx1 = np.maximum(boxes[i, 0], boxes[j, 0])
        yy1 = np.maximum(boxes[i, 1], boxes[j, 1])
        xx2 = np.minimum(boxes[i, 2], boxes[j, 2])
        yy2 = np.minimum(boxes[i, 3], boxes[j, 3])
        inter = np.maximum(xx2 - xx1, np.float32(0)) * np.maximum(yy2 - yy1, np.float32(0))
        iou = inter / (area[i] + area[j] - inter)
        alive[j[iou > np.float32(iou_thres)]] = False
    return np.asarray(keep, dtype=np.int64)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, max_det=300, multi_label=False):
    """prediction: float32 [bs, N, 5 + nc] (xywh, obj, class confidences).  Returns a list of [n, 6] arrays (xyxy, conf, cls) and,
    for the tests, the list of kept candidate indices into N."""
    prediction = np.asarray(prediction, dtype=np.float32)
    bs, _, no = prediction.shape
    nc = no - 5
    out, kept_idx = [], []
    for xi in range(bs):
        x = prediction[xi]
        idx = np.nonzero(x[:, 4] > np.float32(conf_thres))[0]                 # :914, :927
        x = x[idx].copy()
        if x.shape[0] == 0:
            out.append(np.zeros((0, 6), np.float32)); kept_idx.append(np.zeros(0, np.int64))
            continue
        x[:, 5:] *= x[:, 4:5]                                                   # :944
        box = xywh2xyxy(x[:, :4])                                               # :947
        if multi_label and nc > 1:                                               # :921, :951-953: every (box, class) pair above the threshold
            bi, j = np.nonzero(x[:, 5:5 + nc] > np.float32(conf_thres))          # row-major: box ascending, class ascending
            conf = x[bi, 5 + j]
            box, idx = box[bi], idx[bi] * nc + j                                 # candidate id = box * nc + class (the device path's pair index)
        else:
            j = x[:, 5:5 + nc].argmax(1)                                        # :955 (first maximum)
            conf = x[np.arange(x.shape[0]), 5 + j]
        det = np.concatenate((box, conf[:, None], j[:, None].astype(np.float32)), 1)
        m = conf > np.float32(conf_thres)                                       # :956
        if classes is not None:
            m &= np.isin(j, np.asarray(classes))                                # :959-960
        det, idx = det[m], idx[m]
        if det.shape[0] == 0:
            out.append(np.zeros((0, 6), np.float32)); kept_idx.append(np.zeros(0, np.int64))
            continue
        order = np.argsort(-det[:, 4], kind="stable")[:MAX_NMS]                 # :970 (ties: ascending candidate index)
        det, idx = det[order], idx[order]
        c = det[:, 5:6] * np.float32(0 if agnostic else MAX_WH)                 # :973
        keep = nms(det[:, :4] + c, det[:, 4], iou_thres)[:max_det]              # :975-976
        out.append(det[keep]); kept_idx.append(idx[keep])
    return out, kept_idx
