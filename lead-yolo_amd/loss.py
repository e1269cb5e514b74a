"""Target assignment and detection loss (row L of SURVEY.md §8): the reference's
`ComputeLoss.__call__` / `build_targets` (utils/loss.py:121-268) with the EIoU box term
(utils/metrics.py:293-354, incl. its double `+eps` on the union).

Host logic on torch ops, device-agnostic (runs where the predictions live): anchor matching with the
`anchor_t` ratio test, the 5-offset neighbour expansion, `(gxy - offsets).long()` grid indices (int64,
bit-exact with the reference: tests/test_loss.py), BCE objectness with the [4, 1, 0.4] level balance,
`loss * batch_size`.  It allocates its constants once per device and issues no host syncs besides the
data-dependent boolean indexing the algorithm itself requires.
"""
import torch
import torch.nn.functional as F

DEFAULT_HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)


def bbox_eiou(box1, box2, eps=1e-7):
    """EIoU of xywh boxes, row-wise ([n,4] vs [n,4]) -> [n,1]."""
    (x1, y1, w1, h1), (x2, y2, w2, h2) = box1.chunk(4, -1), box2.chunk(4, -1)
    ax1, ax2, ay1, ay2 = x1 - w1 / 2, x1 + w1 / 2, y1 - h1 / 2, y1 + h1 / 2
    bx1, bx2, by1, by2 = x2 - w2 / 2, x2 + w2 / 2, y2 - h2 / 2, y2 + h2 / 2
    inter = (ax2.minimum(bx2) - ax1.maximum(bx1)).clamp(0) * (ay2.minimum(by2) - ay1.maximum(by1)).clamp(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / (union + eps)
    cw = ax2.maximum(bx2) - ax1.minimum(bx1)
    ch = ay2.maximum(by2) - ay1.minimum(by1)
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((bx1 + bx2 - ax1 - ax2) ** 2 + (by1 + by2 - ay1 - ay2) ** 2) / 4
    rw2 = ((bx2 - bx1) - (ax2 - ax1)) ** 2
    rh2 = ((by2 - by1) - (ay2 - ay1)) ** 2
    return iou - (rho2 / c2 + rw2 / (cw ** 2 + eps) + rh2 / (ch ** 2 + eps))


class ComputeLoss:
    sort_obj_iou = False

    def __init__(self, model, autobalance=False, hyp=None):
        det = model.model[-1] if hasattr(model, "model") else model
        self.hyp = dict(DEFAULT_HYP, **(hyp or getattr(model, "hyp", None) or {}))
        if self.hyp["fl_gamma"] > 0 or autobalance:
            raise NotImplementedError("focal loss / autobalance are not used by the LEAD-YOLO recipe and are not built")
        self.na, self.nc, self.nl = det.na, det.nc, det.nl
        self.anchors = det.anchors
        self.balance = {3: [4.0, 1.0, 0.4]}.get(self.nl, [4.0, 1.0, 0.25, 0.06, 0.02])
        self.cp = 1.0 - 0.5 * self.hyp["label_smoothing"]
        self.cn = 0.5 * self.hyp["label_smoothing"]
        self.gr = 1.0
        self._const = {}

    def _consts(self, device):
        c = self._const.get(device)
        if c is None:
            c = dict(off=torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], device=device).float() * 0.5,
                     cls_pw=torch.tensor([self.hyp["cls_pw"]], device=device),
                     obj_pw=torch.tensor([self.hyp["obj_pw"]], device=device))
            self._const[device] = c
        return c

    def build_targets(self, p, targets):
        """-> tcls, tbox, indices (b, a, gj, gi: int64), anch; one entry per detection level."""
        dev = targets.device
        na, nt = self.na, targets.shape[0]
        tcls, tbox, indices, anch = [], [], [], []
        gain = torch.ones(7, device=dev)
        ai = torch.arange(na, device=dev).float().view(na, 1).repeat(1, nt)
        tg = torch.cat((targets.repeat(na, 1, 1), ai[..., None]), 2)
        g = 0.5
        off = self._consts(dev)["off"]
        for i in range(self.nl):
            anchors, shape = self.anchors[i].to(dev), p[i].shape
            gain[2:6] = torch.tensor(shape, device=dev)[[3, 2, 3, 2]]
            t = tg * gain
            if nt:
                r = t[..., 4:6] / anchors[:, None]
                keep = torch.max(r, 1 / r).max(2)[0] < self.hyp["anchor_t"]
                t = t[keep]
                gxy = t[:, 2:4]
                gxi = gain[[2, 3]] - gxy
                j, k = ((gxy % 1 < g) & (gxy > 1)).T
                l, m = ((gxi % 1 < g) & (gxi > 1)).T
                sel = torch.stack((torch.ones_like(j), j, k, l, m))
                t = t.repeat((5, 1, 1))[sel]
                offsets = (torch.zeros_like(gxy)[None] + off[:, None])[sel]
            else:
                t = tg[0]
                offsets = 0
            bc, gxy, gwh, a = t.chunk(4, 1)
            a, (b, c) = a.long().view(-1), bc.long().T
            gij = (gxy - offsets).long()
            gi, gj = gij.T
            indices.append((b, a, gj.clamp(0, shape[2] - 1), gi.clamp(0, shape[3] - 1)))
            tbox.append(torch.cat((gxy - gij, gwh), 1))
            anch.append(anchors[a])
            tcls.append(c)
        return tcls, tbox, indices, anch

    def __call__(self, p, targets):
        dev = p[0].device
        targets = targets.to(dev)
        if dev.type == "cuda" and self.nc == 1 and all(t.dtype == torch.float32 for t in p):
            return _fused_loss(self, list(p), targets.float().contiguous())
        k = self._consts(dev)
        lcls = torch.zeros(1, device=dev)
        lbox = torch.zeros(1, device=dev)
        lobj = torch.zeros(1, device=dev)
        tcls, tbox, indices, anch = self.build_targets(p, targets)
        for i, pi in enumerate(p):
            b, a, gj, gi = indices[i]
            tobj = torch.zeros(pi.shape[:4], dtype=pi.dtype, device=dev)
            n = b.shape[0]
            if n:
                sel = pi[b, a, gj, gi]
                pxy = sel[:, 0:2].sigmoid() * 2 - 0.5
                pwh = (sel[:, 2:4].sigmoid() * 2) ** 2 * anch[i]
                iou = bbox_eiou(torch.cat((pxy, pwh), 1), tbox[i]).squeeze(-1)
                lbox = lbox + (1.0 - iou).mean()
                iou = iou.detach().clamp(0).type(tobj.dtype)
                if self.gr < 1:
                    iou = (1.0 - self.gr) + self.gr * iou
                tobj[b, a, gj, gi] = iou
                if self.nc > 1:
                    pcls = sel[:, 5:]
                    t = torch.full_like(pcls, self.cn)
                    t[torch.arange(n, device=dev), tcls[i]] = self.cp
                    lcls = lcls + F.binary_cross_entropy_with_logits(pcls, t, pos_weight=k["cls_pw"])
            lobj = lobj + F.binary_cross_entropy_with_logits(pi[..., 4], tobj, pos_weight=k["obj_pw"]) * self.balance[i]
        lbox = lbox * self.hyp["box"]
        lobj = lobj * self.hyp["obj"]
        lcls = lcls * self.hyp["cls"]
        bs = p[0].shape[0]
        return (lbox + lobj + lcls) * bs, torch.cat((lbox, lobj, lcls)).detach()


# --------------------------------------------------------------------------------------------------
# Fused device path (csrc/ly_loss.hip): 3 launches per level for the forward AND the gradient, no host sync.
# The torch formulation above stays as the definition (CPU, nc > 1) and as the API for inspecting `build_targets`.
# --------------------------------------------------------------------------------------------------
class _FusedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cl, targets, *preds):
        from . import capi
        dev = preds[0].device
        nl, na = len(preds), cl.na
        nt = targets.shape[0]
        bs = preds[0].shape[0]
        cells = [int(p.shape[0] * p.shape[1] * p.shape[2] * p.shape[3]) for p in preds]
        k = cl._consts(dev)
        if "cells" not in k or k["cells_key"] != tuple(cells):
            k["cells"] = torch.tensor([float(c) for c in cells], device=dev)
            k["balance_t"] = torch.tensor([float(b) for b in cl.balance[:nl]], device=dev)
            k["cells_key"] = tuple(cells)
        zero = torch.zeros(sum(cells) + 4 * nl, dtype=torch.float32, device=dev)          # tobj of every level + accumulators
        winner = torch.full((sum(cells),), -1, dtype=torch.int32, device=dev)
        ncand = 5 * na * nt
        cand_cell = torch.empty((nl, max(ncand, 1)), dtype=torch.int64, device=dev)
        cand = torch.empty((nl, max(ncand, 1), 5), dtype=torch.float32, device=dev)
        acc = zero[sum(cells):].view(nl, 4)
        out = torch.empty(4, dtype=torch.float32, device=dev)
        dps, off = [], 0
        st = capi.stream_ptr()
        for i, p in enumerate(preds):
            p = p.contiguous()
            dp = torch.zeros_like(p)
            anchors = cl.anchors[i].to(dev).float().contiguous()
            _, _, ny, nx, no = p.shape
            capi.check(capi.lib().ly_loss_level(capi.ptr(p), capi.ptr(dp), capi.ptr(anchors), capi.ptr(targets), bs, na, ny, nx, no, nt,
                                                float(cl.hyp["anchor_t"]), float(cl.hyp["box"]), float(cl.hyp["obj"]), float(cl.balance[i]),
                                                capi.ptr(zero[off:off + cells[i]]), capi.ptr(winner[off:off + cells[i]]), capi.ptr(cand_cell[i]),
                                                capi.ptr(cand[i]), capi.ptr(acc[i]), st), "ly_loss_level")
            dps.append(dp)
            off += cells[i]
        capi.check(capi.lib().ly_loss_finish(capi.ptr(acc), nl, capi.ptr(k["cells"]), capi.ptr(k["balance_t"]), float(cl.hyp["box"]),
                                             float(cl.hyp["obj"]), bs, capi.ptr(out), st), "ly_loss_finish")
        ctx.save_for_backward(*dps)
        ctx.mark_non_differentiable(out)
        return out[:1].clone(), out
    @staticmethod
    def backward(ctx, g_loss, _g_items):
        return (None, None) + tuple(dp * g_loss for dp in ctx.saved_tensors)


def _fused_loss(cl, preds, targets):
    if cl.hyp["fl_gamma"] > 0 or cl.hyp["obj_pw"] != 1.0 or cl.gr != 1.0 or cl.sort_obj_iou:
        raise NotImplementedError("fused loss: focal loss / positive weights / gr != 1 are not built")
    loss, out = _FusedLossFn.apply(cl, targets, *preds)
    return loss, out[1:4].detach()
