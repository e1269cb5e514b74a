"""ops.small_grad_scratch: the deferred small-gradient sums are keyed on the autograd graph task: a backward that raises after
taking a scratch loses its end-of-pass callback with the engine's graph task — the next backward must queue a new one and must not flush
the aborted pass's leftovers.  Host logic only (the rounding launch ly_f64_add is replaced by its definition: target += scratch)."""
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
