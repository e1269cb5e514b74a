"""hipGraph capture of the inference forward (serving mode).

The eval forward of lead-yolo-s is ~70 kernel launches of 5-200 us each; ~13 % of the step are small latency-bound kernels
(SE, CoordAtt pools and MLPs, the Detect tails) and every kernel has a ramp-up and a tail during which most CUs idle.  The
images of a batch are independent in eval mode, so the graph runs the batch as `parts` sub-batches on separate streams: while
one sub-batch sits in a small kernel or a kernel tail, the others keep the matrix pipes busy.  Inside each sub-batch the model
forks its own independent branches (SE attention of RFCBAMConv, Detect levels P3 / P4) onto an auxiliary stream when the
graph has a single sub-batch (modules._overlap; nested forks are not capturable on ROCm 7.2).  Captured once (torch.cuda.graph == hipGraph on ROCm) all of this replays without any host work:
bs=32, 640x640: 1.96 ms eager -> 1.87 ms graph -> ~1.75 ms graph with 4 sub-batches.  Results are bit-identical to the eager
single-stream forward (same kernels, same per-element summation order), which bench.py and the tests check.
Every C-ABI entry point only launches kernels on the current stream (no allocation, no synchronisation): that is what makes
the whole forward capturable."""
import torch


def _pick_parts(bs, dtype=torch.float32):
    """sub-batches that run side by side (tools/parts_bench.py, lead-yolo-s 640 x 640 bs = 32: fp32 1 / 2 / 4 / 8 parts 1.71 / 1.56 / 1.55 / 2.10 ms,
    bf16 1.13 / 1.05 / 1.10 / 1.59 ms — the bf16 kernels are half as long, so two streams already fill their tails and four start to split
    launches below a full wave of blocks)"""
    for p in ((2,) if dtype == torch.bfloat16 else (4, 2)):
        if bs % p == 0 and bs // p >= 4:
            return p
    return 1


class GraphedForward:
    """g = GraphedForward(model, example);  (z, [p3, p4, p5]) = g(x)  replays the captured forward on x (same shape / dtype as
    example).  The returned tensors are the graph's static outputs: consume or copy them before the next call."""

    def __init__(self, model, example, parts=None, warmup=2):
        if model.training:
            raise RuntimeError("GraphedForward captures the inference forward: call model.eval() first")
        self.model = model
        self.x = example.clone()
        bs = example.shape[0]
        self.parts = parts = _pick_parts(bs, example.dtype) if parts is None else parts
        if bs % parts:
            raise ValueError(f"batch {bs} is not divisible into {parts} sub-batches")
        xs = list(self.x.chunk(parts, 0))
        self.streams = [torch.cuda.Stream(device=example.device) for _ in range(parts)]
        cur = torch.cuda.current_stream()
        from . import ops
        keep_parts, ops.CONCURRENT_PARTS = ops.CONCURRENT_PARTS, parts
        try:
            self._warm_and_capture(model, xs, parts, cur, warmup, example)
        finally:
            ops.CONCURRENT_PARTS = keep_parts

    def _warm_and_capture(self, model, xs, parts, cur, warmup, example):
        with torch.no_grad():
            for s, xi in zip(self.streams, xs):                  # warm up off the capture (weight packing caches, allocator pools)
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    for _ in range(warmup):
                        model(xi)
            for s in self.streams:
                cur.wait_stream(s)
            torch.cuda.synchronize(example.device)
            self.graph = torch.cuda.CUDAGraph()
            from . import modules
            # nested forks (capture stream -> sub-batch stream -> auxiliary stream) crash hipGraph capture on ROCm 7.2:
            # with several sub-batch streams the per-layer forks stay off (the sub-batches already fill those gaps)
            keep, modules.FORK_BRANCHES = modules.FORK_BRANCHES, parts == 1
            try:
                self._capture(model, xs, parts)
            finally:
                modules.FORK_BRANCHES = keep

    def _capture(self, model, xs, parts):
        from . import modules
        det = model.model[-1] if hasattr(model, "model") else None
        direct = parts > 1 and isinstance(det, modules.Detect) and not det.export
        if direct:
            # sub-batches write straight into batch slices of the full outputs (no concatenation pass)
            with torch.no_grad():
                z0, p0 = model(xs[0])
            sub = xs[0].shape[0]
            z = torch.empty((sub * parts,) + tuple(z0.shape[1:]), dtype=z0.dtype, device=z0.device)
            ps = [torch.empty((sub * parts,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in p0]
            del z0, p0
        # thread_local: API calls of OTHER host threads (e.g. the RCCL watchdog of an initialised process group polling events)
        # must not invalidate the capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            if parts == 1:                                       # directly on the capture stream (the model forks its branches itself)
                self.out = model(xs[0])
                return
            main = torch.cuda.current_stream()
            outs = []
            for k, (s, xi) in enumerate(zip(self.streams, xs)):
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    if direct:
                        lo, hi = k * sub, (k + 1) * sub
                        det._out = dict(z=z[lo:hi], p=[t[lo:hi] for t in ps])
                    try:
                        outs.append(model(xi))
                    finally:
                        if direct:
                            det._out = None
            for s in self.streams:
                main.wait_stream(s)
            if direct:
                self.out = (z, ps)
            elif isinstance(outs[0], tuple) and len(outs[0]) == 2:               # (z, [p_i])
                self.out = (torch.cat([o[0] for o in outs], 0), [torch.cat([o[1][i] for o in outs], 0) for i in range(len(outs[0][1]))])
            else:                                                                  # export mode: (z,)
                self.out = (torch.cat([o[0] for o in outs], 0),)

    def __call__(self, x=None):
        if x is not None and x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        self.graph.replay()
        return self.out
