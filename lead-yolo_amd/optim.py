"""Fused optimiser step: gradient-norm clipping + SGD-nesterov over the reference's three parameter groups +
zero_grad + ModelEMA update as TWO launches of csrc/ly_optim.hip over a device-resident table of tensors, instead of ~50 foreach
launches plus a Python loop over 319 state tensors (reference train.py:330-341; utils/torch_utils.py:318-346, 404-432).

`FusedSGD` is a `torch.optim.Optimizer`: `param_groups` (lr / weight_decay / momentum read every step, so LR schedulers and the
reference's warm-up loop work unchanged), `state[p]["momentum_buffer"]` (checkpoints interchange with torch.optim.SGD), `step()`.
Learning rates, the EMA decay ramp and the step counter live in a small device array: the step has no host-dependent kernel
argument and can be captured into a hipGraph together with forward and backward (train.GraphedTrainStep)."""
import ctypes

import torch

from . import capi

CHUNK = 4096                 # elements per block, = LY_OPT_CHUNK in csrc/ly_optim.hip


def _pairable(p):
    """p and its `_ly_grad_pair` partner can share one gradient allocation (and do not yet): same-shape contiguous fp32 weights whose
    gradients are unset or plain tensors this optimiser may replace (not views of somebody's bucket)"""
    q = getattr(p, "_ly_grad_pair", None)
    if q is None or not q.requires_grad or q.shape != p.shape or q.device != p.device or q.dtype != torch.float32 or not q.is_contiguous():
        return False
    for t in (p, q):
        g = t.grad
        if g is not None and (g.dtype != torch.float32 or g.shape != t.shape or g._base is not None):
            return False
    return True


def _is_tap_major(g, p):
    """g is a view of p's shape [co, ci, kh, kw] over storage laid out [co][kh][kw][ci]"""
    if g is None or g.dim() != 4 or tuple(g.shape) != tuple(p.shape) or g.is_contiguous():
        return False
    co, ci, kh, kw = p.shape
    return tuple(g.stride()) == (kh * kw * ci, 1, kw * ci, ci)


def _owned(p):
    """the optimiser may choose the gradient's storage: there is none yet, or it is a plain tensor of its own (not a slice of a
    larger buffer such as a ddp.GradReducer bucket)"""
    g = p.grad
    return g is None or _is_tap_major(g, p) or (g.is_contiguous() and g.untyped_storage().nbytes() == g.numel() * g.element_size())


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=0.01, momentum=0.937, weight_decay=0.0, nesterov=True, max_norm=10.0):
        if not nesterov or momentum <= 0:
            raise NotImplementedError("FusedSGD implements the LEAD-YOLO recipe: SGD with momentum and nesterov=True")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=True))
        self.max_norm = max_norm
        self._ema = None
        self._table = None
        self._grad_ptrs = None
        self.grad_norm = None           # device scalar: pre-clip global gradient norm of the last step
        self._zeroed = False
        self._sink = None               # the ops.GradSink this optimiser installed; _writes_at_step: its write counter when step() last zeroed the storage
        self._writes_at_step = -1
        self.grad_scale = 1.0           # gradients are read as g * grad_scale: 1 / world_size when the buckets are all-reduced as SUMs

    # ---- EMA -----------------------------------------------------------------------------------------------------
    def attach_ema(self, ema, model):
        """fold `ema.update(model)` (utils/torch_utils.py:418-427) into the step: every floating entry of the model's state_dict"""
        self._ema = (ema, model)
        self._table = None
        return self

    # ---- table ---------------------------------------------------------------------------------------------------
    def _build(self):
        if len(self.param_groups) > 3:
            raise NotImplementedError("FusedSGD carries three learning rates (the reference's bias / weight / norm groups)")
        dev = None
        entries, keep = [], []
        ema_of = {}
        extra = []
        if self._ema is not None:
            ema, model = self._ema
            msd, esd = model.state_dict(), ema.ema.state_dict()
            pid = {p.data_ptr(): p for g in self.param_groups for p in g["params"]}
            for k, v in esd.items():
                if not v.dtype.is_floating_point:
                    continue
                src = msd[k]
                if v.dtype != torch.float32 or src.dtype != torch.float32 or not v.is_contiguous() or not src.is_contiguous():
                    raise NotImplementedError(f"FusedSGD EMA: {k} must be contiguous float32 on both sides")
                if src.data_ptr() in pid:
                    ema_of[src.data_ptr()] = v
                else:
                    extra.append((src, v))
        for gi, group in enumerate(self.param_groups):
            mom = group["momentum"]
            for p in group["params"]:
                if not p.requires_grad:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise NotImplementedError("FusedSGD needs contiguous float32 CUDA parameters (fp32 master weights)")
                dev = p.device
                taps, cin = 1, 0
                if getattr(p, "_ly_tap_major", False) and p.dim() == 4 and p.shape[2] * p.shape[3] > 1 and _owned(p):
                    # k x k convolution weight: gradient storage [cout][kh][kw][cin] (ly_wgrad adds with contiguous atomics), exposed
                    # to torch as a permuted VIEW of the parameter's shape; the update kernel maps indices (LyOptTensor.taps / cin)
                    co, ci, kh, kw = p.shape
                    taps, cin = kh * kw, ci
                    if not _is_tap_major(p.grad, p):
                        store = torch.zeros((co, kh, kw, ci), dtype=torch.float32, device=p.device)
                        view = store.permute(0, 3, 1, 2)
                        if p.grad is not None:
                            view.copy_(p.grad)
                        p.grad = view
                elif _pairable(p):
                    # two weights whose gradients ONE ly_wgrad launch can write as a stacked [2*cout, cin] matrix (C3_CA's cv1 / cv2:
                    # grad.ConvBnActPair): adjacent halves of one allocation
                    q = p._ly_grad_pair
                    both = torch.zeros((2,) + tuple(p.shape), dtype=torch.float32, device=p.device)
                    for half, t in zip(both, (p, q)):
                        if t.grad is not None:
                            half.copy_(t.grad)
                        t.grad = half
                elif p.grad is None:
                    p.grad = torch.zeros_like(p)              # persistent gradient storage: autograd accumulates in place
                elif not p.grad.is_contiguous() and not _is_tap_major(p.grad, p):
                    p.grad = p.grad.contiguous()
                if _is_tap_major(p.grad, p) and taps == 1:
                    co, ci, kh, kw = p.shape
                    taps, cin = kh * kw, ci
                if p.grad.dtype != torch.float32 or not (p.grad.is_contiguous() or taps > 1):
                    raise NotImplementedError("FusedSGD needs float32 gradients, contiguous or tap-major")
                st = self.state[p]
                if "momentum_buffer" not in st or st["momentum_buffer"] is None:
                    st["momentum_buffer"] = torch.zeros_like(p)
                    st["_fresh"] = True
                e = ema_of.get(p.data_ptr())
                entries.append((p.data_ptr(), p.grad.data_ptr(), st["momentum_buffer"].data_ptr(), e.data_ptr() if e is not None else 0, p.numel(),
                                float(group["weight_decay"]), gi, taps, cin))
                keep += [p.grad, st["momentum_buffer"], e]
                if mom != self.param_groups[0]["momentum"]:
                    raise NotImplementedError("FusedSGD uses one momentum for all groups")
        for src, v in extra:
            entries.append((src.data_ptr(), 0, 0, v.data_ptr(), src.numel(), 0.0, -1, 1, 0))
            keep += [src, v]
        if not entries:
            raise ValueError("FusedSGD: no parameters")
        arr = (capi.LyOptTensor * len(entries))(*[capi.LyOptTensor(*e) for e in entries])
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        blk_t, blk_o = [], []
        for i, e in enumerate(entries):
            for off in range(0, e[4], CHUNK):
                blk_t.append(i)
                blk_o.append(off)
        # a fresh momentum buffer is all zeros, and mom*0 + g is torch's first-step rule (buf = g): no first-step flag is needed, so a table
        # rebuilt after load_state_dict (some buffers loaded, some new) never overwrites loaded buffers
        for g in self.param_groups:
            for p in g["params"]:
                if p in self.state:
                    self.state[p].pop("_fresh", None)
        fresh = False
        ema_obj = self._ema[0] if self._ema else None
        hyper = [0.0, 0.0, 0.0, float(self.param_groups[0]["momentum"]), float(self.max_norm or 0.0),
                 float(ema_obj.decay_base) if ema_obj is not None else -1.0, float(ema_obj.tau) if ema_obj is not None else 1.0,
                 float(ema_obj.updates) if ema_obj is not None else 0.0, 1.0 if fresh else 0.0, float(self.grad_scale)]
        self._table = dict(tab=raw.to(dev), blk_t=torch.tensor(blk_t, dtype=torch.int32, device=dev), blk_o=torch.tensor(blk_o, dtype=torch.int64, device=dev),
                           n_blocks=len(blk_t), ws=torch.zeros(1, dtype=torch.float64, device=dev), hyper=torch.tensor(hyper, dtype=torch.float32, device=dev),
                           keep=keep, lrs=None)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._grad_ptrs = [(p, p.grad.data_ptr(), self.state[p]["momentum_buffer"].data_ptr(), float(g["weight_decay"]))
                           for g in self.param_groups for p in g["params"] if p.requires_grad]
        self._zeroed = False
        # the backward kernels may now add weight / BatchNorm gradients straight into this storage (ops.GradSink): no fresh
        # gradient tensors, no zero fills, no AccumulateGrad launches
        from . import ops
        sink = ops.GradSink()
        sink.targets = {id(p): p.grad for g in self.param_groups for p in g["params"] if p.requires_grad}
        ops.SINK = sink
        self._sink = sink

    def _sync_hyper(self):
        """learning rates follow param_groups (schedulers / warm-up write them); one small H2D copy only when they change"""
        t = self._table
        lrs = tuple(float(g["lr"]) for g in self.param_groups) + (float(self.param_groups[0]["momentum"]), float(self.max_norm or 0.0), float(self.grad_scale))
        if lrs != t["lrs"]:
            n = len(self.param_groups)
            t["hyper"][:n].copy_(torch.tensor(lrs[:n], dtype=torch.float32), non_blocking=True)
            t["hyper"][3:5].copy_(torch.tensor(lrs[n:n + 2], dtype=torch.float32), non_blocking=True)     # momentum (warm-up ramps it, train.py:303-311), max_norm
            t["hyper"][9:10].copy_(torch.tensor(lrs[n + 2:], dtype=torch.float32), non_blocking=True)
            t["lrs"] = lrs

    # ---- the step --------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FusedSGD.step does not take a closure")
        capturing = torch.cuda.is_current_stream_capturing()
        if self._table is None:
            if capturing:
                raise RuntimeError("FusedSGD: run one eager step before capturing (the tensor table is built on the first step)")
            self._build()
        if not capturing:
            wds = {id(p): float(g["weight_decay"]) for g in self.param_groups for p in g["params"]}
            for p, ptr, bptr, wd in self._grad_ptrs:      # gradients, momentum buffers and decays must still be what the table says
                buf = self.state[p].get("momentum_buffer")
                if p.grad is None or p.grad.data_ptr() != ptr or buf is None or buf.data_ptr() != bptr or wds.get(id(p)) != wd:
                    self._build()
                    break
            self._sync_hyper()
        t = self._table
        capi.check(capi.lib().ly_optim_step(capi.ptr(t["tab"]), capi.ptr(t["blk_t"]), capi.ptr(t["blk_o"]), t["n_blocks"], capi.ptr(t["ws"]),
                                            capi.ptr(t["hyper"]), capi.ptr(self.grad_norm), capi.stream_ptr()), "ly_optim_step")
        from . import pack
        pack.touch_weights()              # parameters changed through raw pointers: packed-weight images and caches must refresh
        if self._ema is not None and not capturing:
            self._ema[0].updates += 1
        self._zeroed = True
        self._writes_at_step = self._sink.writes if self._sink is not None else -1

    def load_state_dict(self, state_dict):
        """loaded momentum buffers are new tensors: the device table must be rebuilt around them"""
        super().load_state_dict(state_dict)
        self._table = None

    def mark_dirty(self):
        """a backward wrote gradients behind this object's back (a replayed graph: train.GraphedTrainStep)"""
        self._zeroed = False

    def mark_stepped(self):
        """a replayed optimiser graph has just consumed and zeroed the gradients"""
        self._zeroed = True
        self._writes_at_step = self._sink.writes if self._sink is not None else -1

    def _clean(self):
        """the storage is exactly as step() left it: zeroed, and no backward kernel has been handed a target since (the sink counts them;
        a backward that went through autograd's AccumulateGrad instead — sink replaced or `.grad` re-assigned — counts as dirty too)"""
        from . import ops
        return self._zeroed and self._sink is not None and ops.SINK is self._sink and self._sink.writes == self._writes_at_step

    def zero_grad(self, set_to_none=False):
        """Gradients stay allocated (the table, the backward kernels and a captured graph point at them) and step() leaves them zeroed,
        so the reference loop's `optimizer.step(); optimizer.zero_grad()` costs nothing.  Whenever a backward may have run since the last
        step() — a skipped / NaN step whose gradients are to be discarded, an abandoned micro-batch — it zeroes the persistent storage in
        place (one foreach launch)."""
        if self._table is None:
            super().zero_grad(set_to_none=False)
        elif not self._clean():
            grads = [p.grad for g in self.param_groups for p in g["params"] if p.requires_grad and p.grad is not None]
            if grads:
                torch._foreach_zero_(grads)
            self._zeroed = True
            self._writes_at_step = self._sink.writes if self._sink is not None else -1

    fused = True
