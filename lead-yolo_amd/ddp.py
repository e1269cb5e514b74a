"""Data-parallel gradient exchange for the LEAD-YOLO train step: one process per GPU,
`torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm), one all-reduce of all gradients per
optimiser step, overlapped with backward.

Re-designed for this model family rather than copied from DDP's defaults (the reference just wraps
`DistributedDataParallel`, utils/torch_utils.py:55-63): lead-yolo-s has 186 gradient tensors totalling
12.5 MB (n: 3.3 MB, l: 86.8 MB).  DDP's 25 MB bucket cap collapses n/s into ~2 buckets, i.e. almost no
overlap.  xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring all-reduce of S bytes costs
~2*(7/8)*S/153 GB/s: 0.14 ms for 12.5 MB — bandwidth is irrelevant, launch latency and the position of
the first launch are what matter.  So: buckets of ~2 MB in REVERSE registration order (the order
autograd finishes gradients), gradients are views into flat bucket buffers (no copy-in/copy-out), each
bucket's all-reduce is launched asynchronously by the post-accumulate hook of its last gradient, and
`wait()` before the optimiser step only waits — averaging is folded into a pre-scaled all-reduce, or (defer_average, what
train.train_step selects with optim.FusedSGD) into the optimiser's gradient scale: no division pass at all.  A backward that is
replayed from a hipGraph runs no hooks: begin_marks / end_marks leave an event-record node in the graph where each bucket became
complete and exchange(bucket) is released from there on every replay (train.GraphedTrainStep).

The reducer is model-agnostic; tests/test_ddp_gloo.py drives it with world_size 2 on the gloo backend.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, params, process_group=None, bucket_bytes=2 << 20, first_bucket_bytes=512 << 10, average=True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        self.params = [p for p in params if p.requires_grad]
        order = list(reversed(self.params))                       # backward order ~ reverse registration order
        # two weights whose gradients ONE kernel writes as a stacked matrix (`p._ly_grad_pair = q`, C3_CA's cv1 / cv2) sit next to each other,
        # p first — the layout optim.FusedSGD gives them on one GPU; without it the data-parallel step lost the stacked weight gradient
        mine = {id(p) for p in self.params}
        partner_of = {id(p._ly_grad_pair): p for p in self.params if getattr(p, "_ly_grad_pair", None) is not None and id(p._ly_grad_pair) in mine}
        seen, paired = set(), []
        for p in order:
            if id(p) in seen:
                continue
            first = partner_of.get(id(p), p)                      # p is somebody's partner: emit (that one, p)
            q = getattr(first, "_ly_grad_pair", None)
            group = [first, q] if (q is not None and id(q) in mine and q.shape == first.shape and q.dtype == first.dtype) else [p]
            for t in group:
                seen.add(id(t))
            paired.append(group)
        order = paired
        self.buckets = []                                         # list of dict(params, flat, views)
        # ONE master buffer under all buckets of the first parameter's (dtype, device) — every bucket's flat is a slice of it, in bucket order:
        # reset() is one fill, and exchange_all_sync() one collective (the serial form of the captured step, train.GraphedTrainStep)
        flat_params = [p for g in order for p in g]
        self._master, self._master_off = None, 0
        if flat_params:
            d0, v0 = flat_params[0].dtype, flat_params[0].device
            n_master = sum(p.numel() for p in flat_params if p.dtype == d0 and p.device == v0)
            self._master = torch.zeros(n_master, dtype=d0, device=v0)
        cur, cur_bytes, cap = [], 0, first_bucket_bytes           # small first bucket => earliest possible launch
        for group in order:
            nb = sum(p.numel() * p.element_size() for p in group)
            p = group[0]
            if cur and (cur_bytes + nb > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self._close(cur)
                cur, cur_bytes, cap = [], 0, bucket_bytes
            cur.extend(group)
            cur_bytes += nb
        if cur:
            self._close(cur)
        self._slot = {}                                           # id(param) -> (bucket index, index in bucket)
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b["params"]):
                self._slot[id(p)] = (bi, pi)
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._hooks = []
        self._sync = True
        self._direct = set()
        self._deferred = set()
        self._deferred_seen = set()
        self.defer_average = False        # True: buckets are exchanged as plain SUMs, the consumer divides (optim.FusedSGD.grad_scale = 1/world)
        self.exchange_single = False      # True: a world of one still issues its all-reduces (tests of the launch path on one GPU)
        self._mark_fn = None
        self._mark_order = []
        self.reset()

    def _close(self, plist):
        n = sum(p.numel() for p in plist)
        m = self._master
        if m is not None and plist[0].dtype == m.dtype and plist[0].device == m.device and self._master_off + n <= m.numel():
            flat = m[self._master_off:self._master_off + n]
            self._master_off += n
        else:
            flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
        views, off = [], 0
        for p in plist:
            if getattr(p, "_ly_tap_major", False) and p.dim() == 4 and p.shape[2] * p.shape[3] > 1:
                # k x k convolution weight: the slice is laid out [cout][kh][kw][cin] (what the weight-gradient kernels write with unit
                # stride) and exposed as a permuted view of the parameter's shape; an all-reduce is elementwise, the layout is its own business
                co, ci, kh, kw = p.shape
                v = flat[off:off + p.numel()].view(co, kh, kw, ci).permute(0, 3, 1, 2)
            else:
                v = flat[off:off + p.numel()].view_as(p)
            p.grad = v                                            # gradient_as_bucket_view: autograd accumulates in place
            views.append(v)
            off += p.numel()
        self.buckets.append(dict(params=plist, flat=flat, views=views))

    # ---- per-step protocol ---------------------------------------------------------------------------
    def reset(self):
        """Call at the start of every step (instead of zero_grad(set_to_none=True))."""
        whole = self.master_covers_all()
        if whole:
            self._master.zero_()                                  # one fill for every bucket
        for b in self.buckets:
            if not whole:
                b["flat"].zero_()
            for p, v in zip(b["params"], b["views"]):
                p.grad = v
        self._pending = [len(b["params"]) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._direct = set()
        self._deferred = set()
        self._deferred_seen = set()

    def no_sync(self):
        """Context manager for gradient accumulation (the reference's `accumulate` micro-steps, train.py:303-329): backward
        passes inside it only accumulate into the bucket views; the all-reduce of a bucket is launched by the first backward
        OUTSIDE the context (call reset() once per optimiser step, before the first micro-step)."""
        reducer = self

        class _NoSync:
            def __enter__(self):
                reducer._sync = False

            def __exit__(self, *a):
                reducer._sync = True
                reducer._pending = [0 if done else len(b["params"]) for done, b in zip(reducer._launched, reducer.buckets)]
        return _NoSync()

    def attach(self):
        """Register post-accumulate hooks that launch a bucket's all-reduce when its last gradient lands."""
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        try:                                                      # gradients the HIP backward writes in place (ops.GradSink) do not pass
            from . import ops                                     # through autograd's accumulation: they report here instead
            mine = {id(p) for p in self.params}

            def _sink_cb(p):
                if id(p) in mine:
                    if id(p) in self._deferred_seen:
                        self._deferred_seen.discard(id(p))    # (its empty accumulation hook has already been and gone)
                    else:
                        self._direct.add(id(p))           # autograd may still run this parameter's (empty) accumulation hook: ignore it once
                    self._on_grad(p, direct=True)

            def _defer_cb(p):
                # the gradient is still being accumulated elsewhere (ops.small_grad_scratch) and will be announced through _sink_cb when the
                # backward pass ends: the parameter's (empty) accumulation hook, which runs before that, must not count as its arrival
                if id(p) in mine:
                    self._deferred.add(id(p))
            self._sink_cb, self._defer_cb = _sink_cb, _defer_cb
            ops.GRAD_LISTENERS.append(self._sink_cb)
            ops.GRAD_DEFER_LISTENERS.append(self._defer_cb)
        except ImportError:
            self._sink_cb = None
        return self

    def detach(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if getattr(self, "_sink_cb", None) is not None:
            from . import ops
            if self._sink_cb in ops.GRAD_LISTENERS:
                ops.GRAD_LISTENERS.remove(self._sink_cb)
            if getattr(self, "_defer_cb", None) in ops.GRAD_DEFER_LISTENERS:
                ops.GRAD_DEFER_LISTENERS.remove(self._defer_cb)
            self._sink_cb = self._defer_cb = None

    def _on_grad(self, p, direct=False):
        if not direct and id(p) in self._direct:
            self._direct.discard(id(p))                           # already counted when the backward kernel wrote it in place
            return
        if not direct and id(p) in self._deferred:
            self._deferred.discard(id(p))                         # announced later (end of the backward pass): see _defer_cb
            self._deferred_seen.add(id(p))
            return
        bi, pi = self._slot[id(p)]
        view = self.buckets[bi]["views"][pi]
        if p.grad is not view:
            # autograd replaced the view (first accumulation into a None grad): copy into the bucket
            view.copy_(p.grad)
            p.grad = view
        if not self._sync:
            return                                                # accumulation micro-step: no exchange yet
        if self._launched[bi]:
            # the bucket was averaged and exchanged after an earlier backward of this step: adding a second local gradient to
            # it would leave the ranks with different sums (and re-launching would divide by the world size twice)
            raise RuntimeError("GradReducer: a gradient arrived for a bucket that was already all-reduced in this step — call "
                               "reset() before every optimiser step, and run gradient-accumulation micro-steps under no_sync()")
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        self._launched[bi] = True
        if self._mark_fn is not None:
            # the backward is being captured into a hipGraph: leave a mark (an event-record node) where the bucket became complete;
            # every replay releases the bucket's exchange from there (exchange(), train.GraphedTrainStep)
            self._mark_fn(bi)
            self._mark_order.append(bi)
            return
        self.exchange(bi)

    def exchange(self, bi):
        """launch the all-reduce of bucket bi on the current stream's successor (torch.distributed's collective stream), asynchronously"""
        if self.world == 1 and not (self.exchange_single and dist.is_initialized()):
            return
        flat = self.buckets[bi]["flat"]
        if self.average and not self.defer_average:
            flat.div_(self.world)                                 # pre-scale: SUM of pre-divided = mean, no post pass
        self._works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    # ---- a backward that is captured once and replayed (no hooks run at replay) -------------------------------------
    def begin_marks(self, mark_fn):
        """Call inside the capture, before the backward: buckets completing during this backward call mark_fn(bucket index) instead of
        being exchanged.  Requires reset() state (gradients zero, nothing launched)."""
        self._mark_fn = mark_fn
        self._mark_order = []
        self._pending = [len(b["params"]) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._direct = set()
        self._deferred = set()
        self._deferred_seen = set()

    def end_marks(self):
        """-> (bucket indices in the order they were marked, bucket indices no gradient event completed: exchange them after the graph)"""
        order = list(self._mark_order)
        rest = [bi for bi in range(len(self.buckets)) if bi not in set(order)]
        self._mark_fn = None
        self._mark_order = []
        self.begin_external()
        return order, rest

    def begin_external(self):
        """The gradients of this step were produced outside autograd's hooks (a replayed hipGraph of forward + backward wrote them
        into the bucket views, which the previous optimiser step left zeroed): mark every bucket as complete and not yet exchanged,
        so that wait() launches all of them."""
        self._pending = [len(b["params"]) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._direct = set()
        self._deferred = set()
        self._deferred_seen = set()

    def reduce_now(self):
        """For callers without hooks (or unused parameters): launch every bucket not yet launched."""
        for bi, left in enumerate(self._pending):
            if left > 0 and not self._launched[bi]:
                self._pending[bi] = 0
                self._launch(bi)

    def wait(self):
        """Finish the step's exchange.  Afterwards `p.grad` (the bucket views) holds the MEAN over ranks — unless `defer_average` is set
        (train.train_step sets it only while the consumer is an optim.FusedSGD whose `grad_scale` is 1 / world): then the buckets hold the
        SUM over ranks and the optimiser divides while it reads; any other reader of `p.grad` must divide by `self.world` itself."""
        self.reduce_now()
        self.wait_works()

    def master_covers_all(self):
        """every bucket's flat buffer is a slice of the one master buffer (one dtype, one device: the usual case)"""
        m = self._master
        return m is not None and self._master_off == m.numel() and all(
            b["flat"].device == m.device and b["flat"].dtype == m.dtype and
            m.data_ptr() <= b["flat"].data_ptr() < m.data_ptr() + m.numel() * m.element_size() for b in self.buckets)

    def total_bytes(self):
        return sum(b["flat"].numel() * b["flat"].element_size() for b in self.buckets)

    def exchange_all_sync(self):
        """ONE blocking (async_op=False) all-reduce of every gradient, issued from the CURRENT stream.  With the nccl (RCCL) backend the
        collective itself still runs on the process group's internal stream: c10d makes that stream wait for the current one, and the
        current one wait for the collective — two cross-stream hand-offs, nothing overlapping the backward; with gloo the call blocks the
        host.  The serial form of the captured step's exchange; whether it beats the overlapped form (one hand-off, exchange hidden behind
        the rest of the backward) depends on the world size and the gradient volume, so train.GraphedTrainStep TIMES both at construction
        (`dp_exchange="probe"`) instead of assuming.  Requires master_covers_all()."""
        if self.world == 1 and not (self.exchange_single and dist.is_initialized()):
            return
        if self.average and not self.defer_average:
            self._master.div_(self.world)
        dist.all_reduce(self._master, op=dist.ReduceOp.SUM, group=self.group, async_op=False)

    def wait_works(self):
        """the current stream waits for every exchange launched so far"""
        for w in self._works:
            w.wait()
        self._works = []

    # ---- introspection -------------------------------------------------------------------------------
    def plan(self):
        return [dict(n_tensors=len(b["params"]), bytes=b["flat"].numel() * b["flat"].element_size()) for b in self.buckets]
