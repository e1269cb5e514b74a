"""ops.small_grad_scratch: the deferred small-gradient sums are keyed on the autograd graph task: a backward that raises after
taking a scratch loses its end-of-pass callback with the engine's graph task — the next backward must queue a new one and must not flush
the aborted pass's leftovers.  Host logic only (the rounding launch ly_f64_add is replaced by its definition: target += scratch).
Second half: ops._WgradQueue — deferred weight gradients leave in groups of four / at the end of the pass (launches replaced by a recorder)."""
import pytest
import torch

import lead_yolo_amd  # noqa: F401
from lead_yolo_amd import ops


@pytest.fixture
def host_flush(monkeypatch):
    flushed = []

    def flush():
        items, ops._SmallGrads.pending, ops._SmallGrads.task = ops._SmallGrads.pending, [], -1
        for scr, tgt, prm in items:
            tgt += scr.float()
            flushed.append(prm)
    monkeypatch.setattr(ops, "flush_small_grads", flush)
    ops.small_grads_reset()
    yield flushed
    ops.small_grads_reset()


class _Deferred(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, boom):
        ctx.target, ctx.boom = target, boom
        return x * 2.0

    @staticmethod
    def backward(ctx, g):
        assert ops.small_grads_ok()
        scr = ops.small_grad_scratch(ctx.target, None)
        scr += g.double().sum()
        if ctx.boom:
            raise RuntimeError("backward failed after taking a scratch")
        return g * 2.0, None, None


def test_raising_backward_does_not_strand_later_deferred_gradients(host_flush):
    tgt_a, tgt_b = torch.zeros(1), torch.zeros(1)
    x = torch.ones(3, requires_grad=True)
    with pytest.raises(RuntimeError):
        _Deferred.apply(x, tgt_a, True).sum().backward()
    assert len(ops._SmallGrads.pending) == 1            # the aborted pass's entry: its callback died with the graph task
    _Deferred.apply(x, tgt_b, False).sum().backward()
    assert tgt_b.item() == 3.0                          # the next pass queued its own flush ...
    assert tgt_a.item() == 0.0                          # ... and dropped the stale entry instead of flushing half a gradient
    assert ops._SmallGrads.pending == [] and ops._SmallGrads.task == -1
    _Deferred.apply(x, tgt_b, False).sum().backward()   # and the pass after that works as well
    assert tgt_b.item() == 6.0


def test_one_scratch_per_destination_within_a_pass(host_flush):
    """a parameter deferred twice in one backward (a module applied twice) shares ONE scratch: ly_f64_add's table must not hold two entries
    with the same destination (its blocks would race on `dst[i] +=`)"""
    tgt = torch.zeros(1)
    x = torch.ones(2, requires_grad=True)
    (_Deferred.apply(x, tgt, False).sum() + _Deferred.apply(x, tgt, False).sum()).backward()
    assert tgt.item() == 4.0
    assert len(host_flush) == 1


def test_reset_at_step_begin(host_flush):
    x = torch.ones(2, requires_grad=True)
    with pytest.raises(RuntimeError):
        _Deferred.apply(x, torch.zeros(1), True).sum().backward()
    ops.stats_pool_begin(torch.device("cpu"))
    ops.stats_pool_end()
    assert ops._SmallGrads.pending == []


# ---- ops._WgradQueue: deferred, grouped weight gradients -----------------------------------------------------------------------------------
class _FakeSink:
    def __init__(self, targets):
        self.targets = {i: t for i, t in enumerate(targets)}
    is_target = ops.GradSink.is_target


class _FiveWgrads(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dws, boom):
        ctx.dws, ctx.boom = dws, boom
        return x * 2.0

    @staticmethod
    def backward(ctx, g):
        du = torch.zeros(256, 128, dtype=torch.bfloat16)
        xx = torch.zeros(256, 128, dtype=torch.bfloat16)
        for dw in ctx.dws:
            ops.wgrad(M=256, H=16, W=16, N=128, du=du, lddu=128, x=xx, ldx=128, Hin=16, Win=16, Cin=128, dw=dw, lddw=128)
        if ctx.boom:
            raise RuntimeError("backward failed with weight gradients queued")
        return g * 2.0, None, None


def test_deferred_weight_gradients_leave_in_groups_of_four_and_at_the_end_of_the_pass(monkeypatch):
    """plain-row problems of the 128-tile class with sink-resident destinations wait in ops._WgradQueue: four leave as one grouped launch, the
    rest when the backward pass ends; a pass that raises leaves nothing behind for the next one; narrow problems and non-sink destinations
    launch where they are"""
    launched = []
    real_group, real_wgrad = ops.wgrad_group, ops.wgrad

    def group(problems, _now=False):
        if _now:
            launched.append(("group", [id(q["dw"]) for q in problems]))
            return
        return real_group(problems, _now=_now)

    def single(**q):
        if q.get("_now") or not ops._wgrad_deferrable({k: v for k, v in q.items() if k != "_now"}):
            launched.append(("single", id(q["dw"])))
            return
        return real_wgrad(**q)
    dws = [torch.zeros(128, 128) for _ in range(5)]
    monkeypatch.setattr(ops, "SINK", _FakeSink(dws))
    monkeypatch.setattr(ops, "wgrad_group", group)
    monkeypatch.setattr(ops, "wgrad", single)
    monkeypatch.setattr(ops, "WGRAD_DEFER", True)
    ops.small_grads_reset()
    x = torch.ones(3, requires_grad=True)
    _FiveWgrads.apply(x, dws, False).sum().backward()
    assert launched == [("group", [id(t) for t in dws[:4]]), ("single", id(dws[4]))]
    assert ops._WgradQueue.items == [] and ops._WgradQueue.task == -1
    # a pass that raises: its queue is dropped by the next pass (and by small_grads_reset), never launched into another pass's groups
    launched.clear()
    with pytest.raises(RuntimeError):
        _FiveWgrads.apply(x, dws[:2], True).sum().backward()
    assert launched == [] and len(ops._WgradQueue.items) == 2
    _FiveWgrads.apply(x, dws[:1], False).sum().backward()
    assert launched == [("single", id(dws[0]))]
    # not deferrable: destination outside the sink (also: not inside a backward pass)
    q = dict(M=256, H=16, W=16, N=128, du=torch.zeros(256, 128, dtype=torch.bfloat16), lddu=128, x=torch.zeros(256, 128, dtype=torch.bfloat16), ldx=128,
             Hin=16, Win=16, Cin=128, dw=torch.zeros(128, 128), lddw=128)
    assert not ops._wgrad_deferrable(q)
    ops.small_grads_reset()


class _WgradThenDone(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, prms, dws):
        ctx.prms, ctx.dws = prms, dws
        return x * 2.0

    @staticmethod
    def backward(ctx, g):
        du = torch.zeros(256, 128, dtype=torch.bfloat16)
        for prm, dw in zip(ctx.prms, ctx.dws):
            ops.wgrad(M=256, H=16, W=16, N=128, du=du, lddu=128, x=du, ldx=128, Hin=16, Win=16, Cin=128, dw=dw, lddw=128)
            ops.grad_done(prm)
        return g * 2.0, None, None


def test_grad_done_of_a_waiting_weight_gradient_reaches_the_listeners_after_its_launch(monkeypatch):
    """with a gradient listener installed (ddp.GradReducer) the queue stays on: `grad_done(param)` of a problem that still waits is held back
    (the defer listeners are told) and delivered right after the group that contains it has been launched — never before"""
    events = []
    prms = [torch.nn.Parameter(torch.zeros(1)) for _ in range(5)]
    dws = [torch.zeros(128, 128) for _ in range(5)]
    sink = _FakeSink([])
    sink.targets = {id(p): t for p, t in zip(prms, dws)}
    name = {id(p): i for i, p in enumerate(prms)}
    monkeypatch.setattr(ops, "SINK", sink)
    monkeypatch.setattr(ops, "WGRAD_DEFER", True)
    monkeypatch.setattr(ops, "GRAD_LISTENERS", [lambda p: events.append(("done", name[id(p)]))])
    monkeypatch.setattr(ops, "GRAD_DEFER_LISTENERS", [lambda p: events.append(("later", name[id(p)]))])
    monkeypatch.setattr(ops, "wgrad_group", lambda problems, _now=False: events.append(("launch", len(problems))) if _now else None)
    real = ops.wgrad

    def single(**q):
        if q.get("_now"):
            events.append(("launch", 1))
            return
        return real(**q)
    monkeypatch.setattr(ops, "wgrad", single)
    ops.small_grads_reset()
    x = torch.ones(3, requires_grad=True)
    _WgradThenDone.apply(x, prms, dws).sum().backward()
    # under a listener the queue leaves in PAIRS (a waiting gradient holds back its bucket's exchange): problem 0 waits when its grad_done
    # arrives; the second fills the pair (launched inside its own ops.wgrad call), so its own grad_done is immediate; ...; the fifth leaves
    # alone at the end of the pass
    assert events == [("later", 0), ("launch", 2), ("done", 0), ("done", 1), ("later", 2), ("launch", 2), ("done", 2), ("done", 3),
                      ("later", 4), ("launch", 1), ("done", 4)]
    assert ops._WgradQueue.items == [] and ops._WgradQueue.notify == []
    ops.small_grads_reset()


# ---- early flush of completed small-gradient scratches under a gradient listener ----------------------------------------------------------
class _TakeScratches(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, targets, params, log):
        ctx.targets, ctx.params, ctx.log = targets, params, log
        return x * 2.0

    @staticmethod
    def backward(ctx, g):
        scrs = [ops.small_grad_scratch(t, p) for t, p in zip(ctx.targets, ctx.params)]
        ctx.log.append(("taken", len(scrs), [p_.item() for p_ in ctx.targets]))      # what the targets hold when this node has taken its scratches
        for s_ in scrs:
            s_ += 1.0                                                                 # (the node's kernel fills them AFTER taking them all)
        return g * 2.0, None, None, None


def test_completed_small_gradients_leave_early_under_a_listener(monkeypatch):
    """with a gradient listener installed, scratches taken by EARLIER autograd nodes leave as soon as SMALL_GRADS_FLUSH_MIN have collected — never
    the running node's own (it fills them after taking them); without a listener everything leaves at the end of the pass"""
    def host_items(items):
        for scr, tgt, prm in items:
            tgt += scr.float()
        done = set()
        for _, _, prm in items:
            if prm is not None and id(prm) not in done:
                done.add(id(prm))
                ops._SmallGrads.announced.add(id(prm))
                ops.grad_done(prm)
    monkeypatch.setattr(ops, "_flush_small_items", host_items)
    monkeypatch.setattr(ops, "zeros_f64", lambda n, dev: torch.zeros(n, dtype=torch.float64))
    announced = []
    monkeypatch.setattr(ops, "GRAD_LISTENERS", [lambda p: announced.append(id(p))])
    monkeypatch.setattr(ops, "SMALL_GRADS_FLUSH_MIN", 3)
    ops.small_grads_reset()
    ta, tb = [torch.zeros(1) for _ in range(4)], [torch.zeros(1) for _ in range(2)]
    pa, pb = [torch.nn.Parameter(torch.zeros(1)) for _ in range(4)], [torch.nn.Parameter(torch.zeros(1)) for _ in range(2)]
    log = []
    x = torch.ones(3, requires_grad=True)
    # backward order: the outer node (A: 4 scratches) runs first, then the inner node (B: 2)
    _TakeScratches.apply(_TakeScratches.apply(x, tb, pb, log), ta, pa, log).sum().backward()
    # when B had taken its scratches, A's four sums had already been added to their targets (A is an earlier node; its own scratches were NOT
    # flushed while it was still taking them: all four hold the full value)
    assert log[0][:2] == ("taken", 4) and log[1][:2] == ("taken", 2)
    assert [t.item() for t in ta] == [1.0] * 4 and [t.item() for t in tb] == [1.0] * 2
    assert announced[:4] == [id(p) for p in pa] and sorted(announced[4:]) == sorted(id(p) for p in pb)
    # a parameter announced early must not come back in the same pass
    ops.small_grads_reset()
    with pytest.raises(RuntimeError, match="already announced"):          # A takes pa, B takes pb (A's leave), C asks for pa[0] again
        _TakeScratches.apply(_TakeScratches.apply(_TakeScratches.apply(x, [ta[0]], [pa[0]], log), tb, pb, log), ta, pa, log).sum().backward()
    ops.small_grads_reset()
