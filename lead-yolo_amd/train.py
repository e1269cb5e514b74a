"""The reference's optimisation step (train.py:295-341) on the HIP model: SURVEY.md §8 row T.

    imgs.float()/255 -> forward (train-mode, HIP) -> ComputeLoss -> backward (HIP, grad.py) ->
    [gradient all-reduce, ddp.GradReducer] -> clip_grad_norm_(10) -> SGD(nesterov) -> [EMA]

Mirrors `smart_optimizer` (utils/torch_utils.py:318-346: three parameter groups — biases without decay,
BatchNorm weights without decay, all other weights with decay) and `ModelEMA` (utils/torch_utils.py:404-432).
The optimiser is optim.FusedSGD on the GPU (clip + SGD-nesterov + zero_grad + EMA for every tensor in three launches, csrc/ly_optim.hip);
`smart_optimizer(..., fused=False)` returns the reference's torch.optim.SGD object.
"""
import ctypes
import math
import os
import sys
from copy import deepcopy

import torch
from . import ops
import torch.nn as nn

_NORMS = tuple(v for k, v in nn.__dict__.items() if "Norm" in k and isinstance(v, type))


def param_groups(model):
    """-> (biases, decayed weights, norm weights), each in module-traversal order."""
    bias, decay, norm = [], [], []
    for mod in model.modules():
        for name, p in mod.named_parameters(recurse=False):
            if name == "bias":
                bias.append(p)
            elif name == "weight" and isinstance(mod, _NORMS):
                norm.append(p)
            else:
                decay.append(p)
    return bias, decay, norm


def smart_optimizer(model, name="SGD", lr=0.001, momentum=0.9, decay=1e-5, fused=None, max_norm=10.0):
    """The reference's three groups (utils/torch_utils.py:318-346).  fused=True (default when the parameters live on the GPU):
    optim.FusedSGD — clip + SGD-nesterov + zero_grad (+ EMA) in two launches; fused=False: torch.optim.SGD, exactly the
    reference's object."""
    bias, dec, norm = param_groups(model)
    if name != "SGD":
        raise NotImplementedError("the LEAD-YOLO recipe trains with SGD(nesterov); other optimisers are not wired")
    if fused is None:
        fused = all(p.is_cuda for p in bias + dec + norm)
    if fused:
        from .optim import FusedSGD
        opt = FusedSGD(bias, lr=lr, momentum=momentum, nesterov=True, max_norm=max_norm)
    else:
        opt = torch.optim.SGD(bias, lr=lr, momentum=momentum, nesterov=True)
    opt.add_param_group({"params": dec, "weight_decay": decay})
    opt.add_param_group({"params": norm, "weight_decay": 0.0})
    return opt


class ModelEMA:
    """Exponential moving average of the model state (parameters and buffers), decay d*(1 - exp(-updates/tau))."""

    def __init__(self, model, decay=0.9999, tau=2000, updates=0):
        self.ema = deepcopy(model).eval()
        self.updates = updates
        self.decay_base, self.tau = decay, tau
        self.decay = lambda x: decay * (1 - math.exp(-x / tau))
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def update(self, model):
        self.updates += 1
        d = self.decay(self.updates)
        msd = model.state_dict()
        with torch.no_grad():
            for k, v in self.ema.state_dict().items():
                if v.dtype.is_floating_point:
                    v.mul_(d).add_(msd[k].detach(), alpha=1 - d)


def forward_backward(model, compute_loss, imgs, targets, world_size=1, amp=None):
    """uint8 / float batch -> train-mode forward -> loss -> backward; gradients land in `.grad` (train.py:295-324)"""
    if imgs.dtype == torch.uint8 and not getattr(model, "u8_input", False):
        imgs = imgs.float() / 255           # (a model whose first layer is the HIP patch embedding takes the uint8 batch as it is)
    ops.stats_pool_begin(imgs.device)       # one zero fill for all BatchNorm accumulators of the step (ops._StatsPool)
    try:
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            pred = model(imgs)
            loss, items = compute_loss(pred, targets)
        if world_size > 1:
            loss = loss * world_size            # the reducer averages gradients (train.py:321-322)
        loss.backward()
    finally:
        ops.stats_pool_end()
    return loss.detach(), items


def optimizer_step(model, optimizer, ema=None, max_norm=10.0, reducer=None):
    """clip + SGD-nesterov + zero_grad (+ EMA): two launches with optim.FusedSGD, the reference's torch calls otherwise"""
    if getattr(optimizer, "fused", False):
        if ema is not None and (optimizer._ema is None or optimizer._ema[0] is not ema):
            optimizer.attach_ema(ema, model)
        optimizer.max_norm = max_norm
        optimizer.step()
    else:
        assert reducer is None or not reducer.defer_average, "a non-fused optimiser reads p.grad as it is: the reducer must average its buckets"
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm)
        optimizer.step()
        if reducer is None:
            optimizer.zero_grad(set_to_none=True)      # with a reducer the gradients are bucket views, zeroed by reset()
        if ema is not None:
            ema.update(model)


def _set_deferred_average(optimizer, reducer):
    """with the fused optimiser the mean over ranks costs nothing: buckets are all-reduced as SUMs and FusedSGD reads every gradient as
    g / world (one device scalar) — instead of one division launch per bucket.
    Every step re-derives both flags from the pair it is given, so neither sticks: an optimiser later used WITHOUT the reducer reads plain
    gradients again, and a reducer later paired with a non-fused optimiser (torch.optim.SGD, gradient logging, clip_grad_norm_) goes back
    to averaging its buckets itself — `p.grad` is the SUM over ranks only while a FusedSGD with grad_scale = 1 / world is the consumer."""
    fused = getattr(optimizer, "fused", False)
    if reducer is None:
        if fused and optimizer.grad_scale != 1.0:
            optimizer.grad_scale = 1.0
        return
    if fused and reducer.average:
        reducer.defer_average = True
        optimizer.grad_scale = 1.0 / reducer.world
    else:
        reducer.defer_average = False
        if fused:
            optimizer.grad_scale = 1.0


def train_step(model, compute_loss, optimizer, imgs, targets, ema=None, reducer=None, world_size=1, max_norm=10.0, amp=None):
    """One optimisation step; imgs uint8 or float [B,3,H,W] on the model's device, targets [n,6].
    amp: None (fp32 storage) or torch.bfloat16 — forward and loss run inside torch.autocast(dtype=amp), the reference's
    `with torch.cuda.amp.autocast(amp)` region (train.py:316) with bf16 in place of fp16 (no GradScaler needed: bf16 keeps
    fp32's exponent range).  Returns (loss, loss_items) as detached device tensors (no host sync)."""
    _set_deferred_average(optimizer, reducer)
    if reducer is not None:
        reducer.reset()
    loss, items = forward_backward(model, compute_loss, imgs, targets, world_size=world_size, amp=amp)
    if reducer is not None:
        reducer.wait()
    optimizer_step(model, optimizer, ema=ema, max_norm=max_norm, reducer=reducer)
    return loss, items


# The captured data-parallel step can exchange its gradients in two forms (GraphedTrainStep, `dp_exchange`):
#   "overlapped": every ~2 MB bucket's all-reduce is released by an event-record node inside the replayed backward graph and runs on a
#                 communication stream while the rest of the backward executes (what the reference's DistributedDataParallel does,
#                 utils/torch_utils.py:55-63) — costs one cross-stream hand-off per step;
#   "serial":     graph A -> ONE all-reduce of the reducer's master buffer issued from the step's stream -> graph B; nothing overlaps.
#   "probe" (default): both forms are TIMED at construction, at the actual world size over the actual process group, and the faster one is
#                 kept; the decision is taken from the maximum over ranks, so every rank takes the same one.
DP_EXCHANGE = "probe"             # module default of GraphedTrainStep(dp_exchange=None); tests / tools monkeypatch it
DP_PROBE_REPLAYS = 10             # timed replays per form
SPLIT_GRAPHS = False              # development: capture the single-GPU step as the data-parallel pair of graphs (tests monkeypatch it)
DP_MARK_EVERY = 1                 # development: an event node only at every k-th completed bucket


class GraphedTrainStep:
    """The whole optimisation step — uint8 batch -> forward -> loss -> backward -> clip + SGD-nesterov + zero_grad (+ EMA) — captured
    once into hipGraphs and replayed with no per-launch host work.  Possible because every C-ABI entry point only
    launches on the current stream, the device loss has no host sync, and optim.FusedSGD keeps its step-dependent scalars (learning
    rates, EMA ramp, step counter, gradient scale) in device memory.

        step = GraphedTrainStep(model, compute_loss, optimizer, imgs, targets, ema=ema, amp=torch.bfloat16)
        loss, items = step(next_imgs, next_targets)        # same shapes / dtypes; pad `targets` with rows whose image index is -1

    Construction runs `warmup` REAL optimisation steps on the given batch (they size the allocator pools, the statistics pool and
    the optimiser's tensor table) and then captures one more; each call replays it.

    One GPU, accumulate = 1: ONE graph holds the whole step.

    Data parallel (reducer = ddp.GradReducer over RCCL; reference: DistributedDataParallel's reducer, utils/torch_utils.py:55-63,
    train.py:233-235) — two exchange forms, both timed at construction over the actual process group (`dp_exchange="probe"`, the default;
    `.dp_probe` holds both timings and the choice; "serial" / "overlapped" force one, and the ranks are checked to agree).  Overlapped:
    the gradient exchange OVERLAPS the replayed backward: graph A = forward + backward, in which every gradient
    bucket's completion point is an event-record node (csrc ly_event_record); right after launching A the host queues, per bucket in
    completion order, `wait for its event` + `all_reduce(bucket)` on a communication stream, so RCCL starts on a bucket while the graph
    is still executing the rest of the backward; graph B = the fused optimiser (which also divides by the world size:
    FusedSGD.grad_scale — no division pass per bucket) runs when the last exchange is done.  The ~10 all-reduce launches are the only
    eager work of a step.  Serial: graph A -> one all-reduce of the reducer's master buffer issued from the step's stream -> graph B.

    accumulate = k (the reference's gradient accumulation, train.py:157,300,330): call k times with k micro-batches; graph A runs every
    time (gradients add up in place), the exchange and graph B only on every k-th call, which returns `stepped = True` in `.stepped`."""

    def __init__(self, model, compute_loss, optimizer, imgs, targets, ema=None, amp=None, max_norm=10.0, warmup=3, reducer=None, world_size=1,
                 accumulate=1, dp_exchange=None):
        if not getattr(optimizer, "fused", False):
            raise NotImplementedError("GraphedTrainStep needs optim.FusedSGD (smart_optimizer(..., fused=True)): torch.optim.SGD + "
                                      "clip_grad_norm_ keep per-step host state")
        from . import capi, pack
        self.imgs, self.targets = imgs.clone(), targets.clone()
        self.model, self.optimizer, self.ema, self.reducer, self.max_norm = model, optimizer, ema, reducer, max_norm
        self.accumulate, self._micro, self.stepped = max(int(accumulate), 1), 0, False
        self.probe = None
        self.dp_exchange = dp_exchange or DP_EXCHANGE
        if self.dp_exchange not in ("probe", "serial", "overlapped"):
            raise ValueError(f"GraphedTrainStep: dp_exchange must be 'probe', 'serial' or 'overlapped' (got {self.dp_exchange!r})")
        self.dp_probe = None                   # dict(serial_ms, overlapped_ms, chosen, ranks, replays) once both forms were timed
        self._serial = False
        args = dict(ema=ema, amp=amp, max_norm=max_norm, reducer=reducer, world_size=world_size)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=imgs.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for i in range(max(warmup, 1)):
                if i == max(warmup, 1) - 1:
                    pack.PLAN.trace = set()    # which packed weight images this step asks for: the capture refreshes exactly those
                try:
                    train_step(model, compute_loss, optimizer, self.imgs, self.targets, **args)
                finally:
                    traced, pack.PLAN.trace = pack.PLAN.trace, None
                    ops.small_grads_reset()           # (a warm-up step that raised must not leave deferred gradients pending)
        cur.wait_stream(side)
        torch.cuda.synchronize(imgs.device)
        self.graph = torch.cuda.CUDAGraph()
        self.opt_graph = None
        self._events, self._marked, self._unmarked, self._comm = [], [], [], None
        # the capture launches a PRIVATE descriptor table over the traced images: the global one is rebuilt (its tensors freed) whenever any
        # model of the process registers a new image or dies, and a replay addressing it read freed memory (intermittent GPU fault)
        self._pack_table, self._pack_images = pack.PLAN.private_table(sorted(traced or (), key=repr))
        pack.PLAN.capture_table = self._pack_table
        pack.touch_weights()                   # the captured step must begin with the (single) refresh of every packed weight image
        try:
            self._capture(model, compute_loss, optimizer, ema, amp, max_norm, reducer, world_size, capi)
        finally:
            # also on a failed capture (bench.py falls back to eager): the global plan must not keep launching this object's private table,
            # and the reducer must not keep calling the capture's marker
            pack.PLAN.capture_table = None
            if reducer is not None and getattr(reducer, "_mark_fn", None) is not None:
                reducer._mark_fn, reducer._mark_order = None, []
                reducer.reset()
        # Everything the graphs address through raw pointers must outlive them: the step's zero pool (ops._POOL.buf is replaced
        # when a later, larger step grows it), the loss constants (replaced when the level shapes change), the optimiser's tensor table
        # (rebuilt when a gradient pointer changes).  A later eager step or a second GraphedTrainStep at another shape would otherwise
        # free memory these graphs still zero-fill and add into.
        self._pins = (ops._POOL.buf, dict(getattr(compute_loss, "_const", {})), optimizer._table)
        # packed weight images were only RECORDED as refreshed during the capture: an eager forward before the first replay must rebuild them
        pack.touch_weights()
        if reducer is not None and self.accumulate == 1:
            self._choose_exchange()

    def _capture(self, model, compute_loss, optimizer, ema, amp, max_norm, reducer, world_size, capi):
        imgs = self.imgs
        if reducer is None and self.accumulate == 1 and not SPLIT_GRAPHS:
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.loss, self.items = train_step(model, compute_loss, optimizer, self.imgs, self.targets, ema=ema, amp=amp, max_norm=max_norm)
        else:
            lib = capi.lib()
            mark = None
            if reducer is not None:
                # graph A always carries the bucket-completion event nodes (they cost nothing inside a replayed graph: DESIGN.md §6); which
                # exchange form uses them is decided after the capture (_choose_exchange)
                _set_deferred_average(optimizer, reducer)
                reducer.reset()
                for _ in reducer.buckets:
                    ev = capi._P()
                    capi.check(lib.ly_event_create(ctypes.byref(ev)), "ly_event_create")
                    self._events.append(ev)
                self._comm = torch.cuda.Stream(device=imgs.device)

                self._recorded = set()
                stride = max(int(DP_MARK_EVERY), 1)

                def mark(bi):
                    # runs inside the captured backward (autograd's thread, capture stream current): an event-record node behind everything
                    # captured so far, i.e. behind the kernel that wrote the bucket's last gradient.  (DP_MARK_EVERY = k: a node only at
                    # every k-th completed bucket — the buckets in between are released by the next node or after the graph)
                    if len(reducer._mark_order) % stride == stride - 1:
                        capi.check(lib.ly_event_record(self._events[bi], capi.stream_ptr()), "ly_event_record")
                        self._recorded.add(bi)
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                if reducer is not None:
                    reducer.begin_marks(mark)
                self.loss, self.items = forward_backward(model, compute_loss, self.imgs, self.targets, world_size=world_size, amp=amp)
                if reducer is not None:
                    self._marked, self._unmarked = reducer.end_marks()
            self.opt_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.opt_graph, pool=self.graph.pool(), capture_error_mode="thread_local"):
                optimizer_step(model, optimizer, ema=ema, max_norm=max_norm, reducer=reducer)

    # ---- which exchange form: measured, at the actual world size -----------------------------------------------------------------
    def _dp_state(self):
        """every tensor an optimisation step changes (parameters, BatchNorm buffers, momentum, EMA, the optimiser's device scalars): the probe
        replays real steps and puts them back"""
        ts = [v for v in self.model.state_dict().values()]
        ts += [st["momentum_buffer"] for g in self.optimizer.param_groups for p in g["params"]
               for st in (self.optimizer.state.get(p, {}),) if st.get("momentum_buffer") is not None]
        if self.ema is not None:
            ts += list(self.ema.ema.state_dict().values())
        t = self.optimizer._table
        ts += [t["hyper"], t["ws"], self.optimizer.grad_norm]
        return ts

    def _choose_exchange(self):
        import torch.distributed as dist
        from . import pack
        red = self.reducer
        exchanging = not (red.world == 1 and not (red.exchange_single and dist.is_initialized()))
        can_serial = red.master_covers_all()
        want = self.dp_exchange
        if want == "serial" and not can_serial:
            raise RuntimeError("GraphedTrainStep: dp_exchange='serial' needs every gradient bucket inside the reducer's master buffer (one dtype, one device)")
        if exchanging and dist.is_initialized():
            # the ranks must agree on the form (they issue different collective sequences): compare the request across the group first
            code = {"probe": 0, "serial": 1, "overlapped": 2}[want] if can_serial else 3
            dev = self.imgs.device if dist.get_backend(red.group) == "nccl" else torch.device("cpu")
            lo, hi = torch.tensor([code], device=dev), torch.tensor([code], device=dev)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=red.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=red.group)
            if int(lo) != int(hi):
                raise RuntimeError(f"GraphedTrainStep: the ranks disagree on dp_exchange (codes {int(lo)} .. {int(hi)}): every rank must request the same form")
        if want != "probe" or not can_serial or not exchanging:
            self._serial = want == "serial"          # (nothing to exchange / nothing to choose from: the overlapped form unless asked otherwise)
            return
        saved = [t.clone() for t in self._dp_state()]
        updates = self.ema.updates if self.ema is not None else 0
        times = {}
        dev = self.imgs.device
        for form in ("serial", "overlapped"):
            self._serial = form == "serial"
            for _ in range(2):
                self.__call__()
            torch.cuda.synchronize(dev)
            if dist.is_initialized():
                dist.barrier(group=red.group)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(DP_PROBE_REPLAYS):
                self.__call__()
            e1.record()
            torch.cuda.synchronize(dev)
            times[form] = e0.elapsed_time(e1) / DP_PROBE_REPLAYS
        t = torch.tensor([times["serial"], times["overlapped"]], dtype=torch.float64,
                         device=dev if (dist.is_initialized() and dist.get_backend(red.group) == "nccl") else torch.device("cpu"))
        if dist.is_initialized():
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=red.group)      # the slowest rank sets the step time; identical numbers everywhere
        ser, ovl = float(t[0]), float(t[1])
        self._serial = ser < ovl
        self.dp_probe = dict(serial_ms=round(ser, 4), overlapped_ms=round(ovl, 4), chosen="serial" if self._serial else "overlapped",
                             ranks=red.world, replays=DP_PROBE_REPLAYS)
        with torch.no_grad():
            for dst, src in zip(self._dp_state(), saved):
                dst.copy_(src)
        if self.ema is not None:
            self.ema.updates = updates
        pack.touch_weights()
        torch.cuda.synchronize(dev)

    def __del__(self):
        # the bucket events are released only while the interpreter (and with it the HIP runtime) is certainly alive: destroying them from a
        # finaliser that runs during shutdown aborted the process after a green test run (1 of 3 runs); a handful of events leaked at exit are harmless
        try:
            if sys is None or sys.is_finalizing():
                return
            from . import capi
            evs, self._events = self._events, []
            for ev in evs:
                capi.lib().ly_event_destroy(ev)
        except Exception:                                   # noqa: BLE001
            pass

    def _load(self, imgs, targets):
        if imgs is not None and imgs.data_ptr() != self.imgs.data_ptr():
            if imgs.dtype != self.imgs.dtype or tuple(imgs.shape) != tuple(self.imgs.shape):
                raise ValueError(f"GraphedTrainStep: the batch must keep the captured dtype and shape {self.imgs.dtype} {tuple(self.imgs.shape)} "
                                 f"(got {imgs.dtype} {tuple(imgs.shape)}); a float [0, 1] batch copied into the captured uint8 buffer would "
                                 "truncate to zeros")
            self.imgs.copy_(imgs, non_blocking=True)
        if targets is not None and targets.data_ptr() != self.targets.data_ptr():
            if tuple(targets.shape) != tuple(self.targets.shape) or targets.dtype != self.targets.dtype:
                raise ValueError(f"GraphedTrainStep: targets must keep the captured shape {tuple(self.targets.shape)} (pad with rows whose "
                                 f"image index is -1), got {tuple(targets.shape)} {targets.dtype}")
            self.targets.copy_(targets, non_blocking=True)

    def profile_step(self):
        """One data-parallel step (accumulate = 1, reducer present) with HIP events around its parts: -> dict(graph_a_ms, step_ms,
        buckets=[(bucket, bytes, released at % of graph A)]).  `released` = when the bucket's all-reduce was queued behind its in-graph
        event on the communication stream, relative to the span of graph A (forward + backward): < 100 means the exchange started while
        the backward was still running.  Introspection for bench.py / tools; the step it runs is a real optimisation step."""
        if self.opt_graph is None or self.reducer is None or self.accumulate != 1:
            raise RuntimeError("profile_step: needs the data-parallel form (reducer, accumulate = 1)")
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        self.probe, self._a_done = [], e1
        e0.record()
        try:
            self.__call__()
        finally:
            self._a_done = None
        e2.record()
        torch.cuda.synchronize()
        probe, self.probe = self.probe, None
        a_ms = e0.elapsed_time(e1)
        return dict(graph_a_ms=a_ms, step_ms=e0.elapsed_time(e2),
                    buckets=[(bi, self.reducer.buckets[bi]["flat"].numel() * self.reducer.buckets[bi]["flat"].element_size(),
                              round(100.0 * e0.elapsed_time(ev) / a_ms, 1)) for bi, ev in probe])

    def __call__(self, imgs=None, targets=None):
        from . import capi, pack
        self._load(imgs, targets)
        _set_deferred_average(self.optimizer, self.reducer)    # an eager train_step in between may have reset the (grad_scale, defer_average) pair
        self.optimizer._sync_hyper()                       # learning-rate schedule -> device (only when it changed)
        if self.opt_graph is None:
            self.graph.replay()
            pack.touch_weights()                           # parameters / running statistics changed behind torch's version counters
            if self.ema is not None:
                self.ema.updates += 1
            self.stepped = True
            return self.loss, self.items
        self.graph.replay()                                # forward + backward: gradients add into the persistent storage / bucket views
        if getattr(self, "_a_done", None) is not None:
            self._a_done.record()                          # (profile_step)
        self.optimizer.mark_dirty()                        # (a zero_grad() now — an abandoned micro-batch — must really zero them)
        self._micro += 1
        self.stepped = self._micro >= self.accumulate
        if not self.stepped:
            return self.loss, self.items
        self._micro = 0
        if self.reducer is not None and getattr(self, "_serial", False):
            self.reducer.exchange_all_sync()               # one collective on this stream, between the two graphs
        elif self.reducer is not None:
            lib, red, cur, comm = capi.lib(), self.reducer, torch.cuda.current_stream(), self._comm
            with torch.cuda.stream(comm):
                held = []
                for bi in self._marked:                    # released from the middle of the running graph, bucket by bucket
                    held.append(bi)
                    if bi not in self._recorded:
                        continue                           # no node of its own: goes out with the next recorded bucket
                    capi.check(lib.ly_stream_wait_event(capi._P(comm.cuda_stream), self._events[bi]), "ly_stream_wait_event")
                    for bj in held:
                        if self.probe is not None:         # tools/dp_overlap_probe.py: when was the bucket released / its exchange queued
                            ev = torch.cuda.Event(enable_timing=True)
                            ev.record(comm)
                            self.probe.append((bj, ev))
                        red.exchange(bj)
                    held = []
                if self._unmarked or held:                 # buckets no gradient event completed (unused parameters) / behind the last node: after the graph
                    comm.wait_stream(cur)
                    for bi in held + list(self._unmarked):
                        red.exchange(bi)
                works = bool(red._works)
                active = not (red.world == 1 and not (red.exchange_single and torch.distributed.is_initialized()))
            # the step's stream waits for the collectives' work objects DIRECTLY: one cross-stream hand-off at the end of the backward instead of
            # two (communication stream -> step stream on top of RCCL's own; each costs ~0.1 ms).  Everything queued on the communication stream
            # (the event waits, exchange()'s pre-scale) is ordered before the collective that the work object stands for.  Only an exchange
            # that ran but left no work object (never the case with async_op=True) would still need the communication stream itself.
            red.wait_works()
            if active and not works:
                cur.wait_stream(comm)
        self.opt_graph.replay()
        self.optimizer.mark_stepped()
        pack.touch_weights()
        if self.ema is not None:
            self.ema.updates += 1
        return self.loss, self.items
