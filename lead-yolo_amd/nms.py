"""Eval tail on the device: the host mirror of the reference's `non_max_suppression` (utils/general.py:884-994; called by detect.py:149 and
val.py:230-234 on Detect's inference output).  Same name, arguments and return value — a list with one [n, 6] tensor (xyxy, conf, cls) per
image — for the options the detector's own callers use; the arithmetic runs in csrc/ly_nms.hip (candidate scoring, greedy NMS) around one
torch.sort on the device, with no per-image host loop.  `nms_padded` is the sync-free form (padded rows + counts)."""
import ctypes

import torch

from . import capi

MAX_WH = 7680        # utils/general.py:918
MAX_NMS = 30000      # utils/general.py:919


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def nms_padded(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, max_det=300, multi_label=False):
    """(dets [bs, max_det, 6] float32 — rows past the count are zero —, counts [bs] int32, keep [bs, max_det] int32 indices into the
    prediction rows); no host synchronisation, capturable into a hipGraph."""
    if isinstance(prediction, (list, tuple)):          # model in validation mode: (inference_out, loss_out)  (utils/general.py:904-905)
        prediction = prediction[0]
    if not (0 <= conf_thres <= 1) or not (0 <= iou_thres <= 1):
        raise ValueError(f"invalid thresholds conf={conf_thres} iou={iou_thres}: valid values are between 0.0 and 1.0")
    if not prediction.is_cuda:
        raise RuntimeError(f"non_max_suppression: the HIP path needs a CUDA/ROCm tensor (got {prediction.device}); there is no CPU fallback")
    pred = prediction.float().contiguous()
    bs, n, no = pred.shape
    nc = no - 5
    if nc < 1:
        raise ValueError(f"prediction rows must hold xywh, obj and at least one class confidence (got {no} columns)")
    mask = 0
    if classes is not None:
        for c in classes:
            if not 0 <= int(c) < 64:
                raise NotImplementedError("class filter: class ids must be below 64")
            mask |= 1 << int(c)
        if mask == 0:
            mask = 1 << 63 if nc < 64 else 0          # an empty filter keeps nothing
            if nc >= 64:
                raise NotImplementedError("an empty class filter with 64 or more classes")
    dev = pred.device
    lib, st = capi.lib(), capi.stream_ptr()
    multi_label = bool(multi_label) and nc > 1               # utils/general.py:921
    if multi_label:
        # every (box, class) pair above the threshold is a candidate (val.py's setting for nc > 1): pair index box * nc + class; the returned
        # `keep` indices address these pairs (box = keep // nc)
        n_box, n = n, n * nc
        score = torch.empty((bs, n), dtype=torch.float32, device=dev)
        det = torch.empty((bs, n, 6), dtype=torch.float32, device=dev)
        capi.check(lib.ly_nms_candidates_ml(_p(pred), bs, n_box, no, float(conf_thres), ctypes.c_ulonglong(mask), _p(score), _p(det), st),
                   "ly_nms_candidates_ml")
    else:
        score = torch.empty((bs, n), dtype=torch.float32, device=dev)
        det = torch.empty((bs, n, 6), dtype=torch.float32, device=dev)
        capi.check(lib.ly_nms_candidates(_p(pred), bs, n, no, float(conf_thres), ctypes.c_ulonglong(mask), _p(score), _p(det), st), "ly_nms_candidates")
    svals, order = torch.sort(score, dim=1, descending=True, stable=True)          # x[:, 4].argsort(descending=True)  (utils/general.py:970)
    keep = torch.zeros((bs, max_det), dtype=torch.int32, device=dev)
    count = torch.empty((bs,), dtype=torch.int32, device=dev)
    capi.check(lib.ly_nms_greedy(_p(det), _p(order), _p(svals), bs, n, float(iou_thres), 0.0 if agnostic else float(MAX_WH), int(max_det), MAX_NMS,
                                 _p(keep), _p(count), st), "ly_nms_greedy")
    rows = torch.gather(det, 1, keep.long().unsqueeze(-1).expand(bs, max_det, 6))
    valid = torch.arange(max_det, device=dev).unsqueeze(0) < count.unsqueeze(1)
    return rows * valid.unsqueeze(-1), count, keep


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, labels=(), max_det=300, nm=0):
    """Reference signature (utils/general.py:884-894).  Returns a list of [n, 6] tensors (xyxy, conf, cls), one per image."""
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    nc = prediction.shape[2] - nm - 5
    if nm:
        raise NotImplementedError("mask coefficients (nm > 0) belong to the segmentation models, which are outside the LEAD-YOLO path")
    if labels:
        raise NotImplementedError("a-priori labels (autolabelling) are not built on the device path")
    dets, count, _ = nms_padded(prediction, conf_thres, iou_thres, classes, agnostic, max_det, multi_label=multi_label)
    counts = count.tolist()                            # the list-of-tensors return value needs the lengths on the host: one sync
    return [dets[i, :c] for i, c in enumerate(counts)]
