"""Target assignment and detection loss on the device (row L of SURVEY.md §8): what the reference's
`ComputeLoss.__call__` / `build_targets` (utils/loss.py:121-268) and `bbox_iou(EIoU=True)` (utils/metrics.py:293-354,
including its double `+eps` on the union) compute, done by csrc/ly_loss.hip in 3 launches per level for the forward AND
the gradient, with no host sync.

There is no torch formulation in this package: the CPU restatement used for checking lives in oracle/functional.py
(test infrastructure).  `ComputeLoss` therefore needs CUDA predictions and raises otherwise.  Any nc: the class BCE of
utils/loss.py:168-173 (label smoothing, cls_pw) is part of the kernel when nc > 1 (LEAD-YOLO.yaml itself is nc = 1); obj_pw too.
`build_targets` is the inspectable view of the kernel's anchor matching: it runs the matching kernel alone and reads its candidate buffers back (int64 indices bit-exact with the reference's vectors, tests/test_loss.py).
"""
import torch

from . import capi, ops

DEFAULT_HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)


class ComputeLoss:
    sort_obj_iou = False

    def __init__(self, model, autobalance=False, hyp=None):
        det = model.model[-1] if hasattr(model, "model") else model
        self.hyp = dict(DEFAULT_HYP, **(hyp or getattr(model, "hyp", None) or {}))
        if self.hyp["fl_gamma"] > 0 or autobalance:
            raise NotImplementedError("focal loss / autobalance are not used by the LEAD-YOLO recipe and are not built")
        self.na, self.nc, self.nl = det.na, det.nc, det.nl
        self.cp, self.cn = 1.0 - 0.5 * self.hyp["label_smoothing"], 0.5 * self.hyp["label_smoothing"]      # smooth_BCE (utils/loss.py:17-19)
        self.anchors = det.anchors
        self.balance = {3: [4.0, 1.0, 0.4]}.get(self.nl, [4.0, 1.0, 0.25, 0.06, 0.02])
        self.gr = 1.0
        self._const = {}

    # ---- shared plumbing -----------------------------------------------------------------------------------------
    def _check(self, p, targets):
        if not p[0].is_cuda:
            raise RuntimeError(f"ComputeLoss: the loss runs on the GPU (csrc/ly_loss.hip); got predictions on {p[0].device} — "
                               "there is no CPU fallback (the CPU restatement is oracle/functional.py, test infrastructure)")
        if targets.dim() != 2 or targets.shape[1] != 6:
            raise ValueError(f"ComputeLoss: targets must be [n, 6] (image, class, x, y, w, h), got {tuple(targets.shape)}")
        return targets.to(p[0].device, torch.float32).contiguous()

    def _consts(self, device, cells):
        k = self._const.get(device)
        if k is None or k["cells_key"] != tuple(cells):
            k = dict(cells=torch.tensor([float(c) for c in cells], device=device),
                     balance=torch.tensor([float(b) for b in self.balance[:len(cells)]], device=device),
                     anchors=[self.anchors[i].to(device).float().contiguous() for i in range(self.nl)], cells_key=tuple(cells))
            self._const[device] = k
        return k

    def _levels(self, preds, targets, match_only):
        """launches ly_loss_level for every level; returns the buffers"""
        dev = preds[0].device
        nl, na = len(preds), self.na
        nt = targets.shape[0]
        bs = preds[0].shape[0]
        cells = [int(p.shape[0] * p.shape[1] * p.shape[2] * p.shape[3]) for p in preds]
        k = self._consts(dev, cells)
        ncand = max(5 * na * nt, 1)
        zero = ops.zeros_f32(sum(cells) + 8 * nl, dev)                                       # tobj of every level + accumulators (step pool)
        winner = torch.full((sum(cells),), -1, dtype=torch.int32, device=dev)
        cand_cell = torch.empty((nl, ncand), dtype=torch.int64, device=dev)
        cand = torch.empty((nl, ncand, 5), dtype=torch.float32, device=dev)
        tbox = torch.empty((nl, ncand, 4), dtype=torch.float32, device=dev) if match_only else None
        acc = zero[sum(cells):].view(nl, 8)
        dps, off = [], 0
        st = capi.stream_ptr()
        for i, p in enumerate(preds):
            dp = None if match_only else ops.zeros_f32(p.numel(), dev).view(p.shape)      # read by the backward of the same step only
            _, _, ny, nx, no = p.shape
            capi.check(capi.lib().ly_loss_level(capi.ptr(p), capi.ptr(dp), capi.ptr(k["anchors"][i]), capi.ptr(targets), bs, na, ny, nx, no, nt,
                                                float(self.hyp["anchor_t"]), float(self.hyp["box"]), float(self.hyp["obj"]), float(self.balance[i]),
                                                capi.ptr(zero[off:off + cells[i]]), capi.ptr(winner[off:off + cells[i]]), capi.ptr(cand_cell[i]),
                                                capi.ptr(cand[i]), capi.ptr(acc[i]), capi.ptr(tbox[i]) if match_only else capi.ptr(None),
                                                int(match_only), float(self.hyp["cls"]), float(self.cp), float(self.cn), float(self.hyp["cls_pw"]),
                                                float(self.hyp["obj_pw"]), st), "ly_loss_level")
            dps.append(dp)
            off += cells[i]
        return dict(dps=dps, acc=acc, cand_cell=cand_cell, tbox=tbox, k=k, bs=bs, nt=nt)

    # ---- the reference's API ---------------------------------------------------------------------------------------
    def build_targets(self, p, targets):
        """-> tcls, tbox, indices (b, a, gj, gi: int64), anch; one entry per detection level, rows in the reference's order
        (offset k, anchor a, target t).  Inspection API: reads the matching kernel's buffers back (one host sync)."""
        targets = self._check(p, targets)
        preds = [t.detach().float().contiguous() for t in p]
        r = self._levels(preds, targets, match_only=True)
        if float(r["acc"][:, 3].sum()) > 0:
            raise IndexError("build_targets: a target row has an image index outside the batch (or a NaN)")
        tcls, tbox, indices, anch = [], [], [], []
        nt = r["nt"]
        for i, pi in enumerate(preds):
            _, na, ny, nx, _ = pi.shape
            if nt == 0:                                                         # no labels: every list entry is empty
                z = torch.zeros(0, dtype=torch.int64, device=pi.device)
                indices.append((z, z.clone(), z.clone(), z.clone()))
                tbox.append(torch.zeros(0, 4, device=pi.device))
                anch.append(r["k"]["anchors"][i][z])
                tcls.append(z.clone())
                continue
            cell = r["cand_cell"][i][:5 * na * nt]
            keep = cell >= 0
            cell = cell[keep]
            gi = cell % nx
            gj = (cell // nx) % ny
            a = (cell // (nx * ny)) % na
            b = cell // (nx * ny * na)
            indices.append((b, a, gj, gi))
            tbox.append(r["tbox"][i][:5 * na * nt][keep])
            anch.append(r["k"]["anchors"][i][a])
            tidx = torch.arange(keep.numel(), device=pi.device)[keep] % nt          # candidate index (k*na + a)*nt + t -> target row t
            tcls.append(targets[tidx, 1].long())
        return tcls, tbox, indices, anch

    def __call__(self, p, targets):
        targets = self._check(p, targets)
        if self.gr != 1.0 or self.sort_obj_iou:
            raise NotImplementedError("device loss: gr != 1 / sort_obj_iou are not built")
        loss, out = _LossFn.apply(self, targets, *p)
        return loss, out[1:4].detach()


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cl, targets, *preds):
        ctx.dtypes = [p.dtype for p in preds]
        preds = [p.float().contiguous() for p in preds]                     # bf16 / fp16 heads: the loss itself is fp32 (as under autocast)
        r = cl._levels(preds, targets, match_only=False)
        out = torch.empty(4, dtype=torch.float32, device=preds[0].device)
        capi.check(capi.lib().ly_loss_finish(capi.ptr(r["acc"]), len(preds), capi.ptr(r["k"]["cells"]), capi.ptr(r["k"]["balance"]),
                                             float(cl.hyp["box"]), float(cl.hyp["obj"]), float(cl.hyp["cls"]), int(cl.nc), r["bs"], capi.ptr(out),
                                             capi.stream_ptr()), "ly_loss_finish")
        ctx.save_for_backward(*r["dps"])
        ctx.mark_non_differentiable(out)
        return out[:1].clone(), out

    @staticmethod
    def backward(ctx, g_loss, _g_items):
        dps = list(ctx.saved_tensors)
        if all(dp.is_cuda and dp.dtype == torch.float32 for dp in dps) and g_loss.numel() == 1 and g_loss.dtype == torch.float32:
            scaled = torch._foreach_mul(dps, g_loss.reshape(()))            # the levels' gradients scaled in ONE multi-tensor launch (were three)
        else:
            scaled = [dp * g_loss for dp in dps]
        return (None, None) + tuple(sp.to(dt) for sp, dt in zip(scaled, ctx.dtypes))
