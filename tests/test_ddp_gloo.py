"""world_size-2 gloo test of the gradient reducer (the N>1 path of the train step; RCCL on the GPU box)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lead_yolo_amd.ddp import GradReducer
        torch.manual_seed(0)                                       # identical replicas
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.SiLU(),
                                  torch.nn.Conv2d(8, 16, 1), torch.nn.Flatten(), torch.nn.Linear(16 * 8 * 8, 5))
        red = GradReducer(net.parameters(), bucket_bytes=4096, first_bucket_bytes=256).attach()
        plan = red.plan()
        assert sum(b["n_tensors"] for b in plan) == len(list(net.parameters()))
        assert len(plan) >= 3                                      # several buckets => overlap is possible
        ref_grads = None
        for step in range(2):
            torch.manual_seed(100 + rank + 10 * step)              # different shard per rank
            x = torch.randn(4, 3, 8, 8)
            y = torch.randn(4, 5)
            # expected: mean over ranks of the local gradients
            local = torch.autograd.grad(((net(x) - y) ** 2).mean(), list(net.parameters()))
            gathered = []
            for g in local:
                buf = [torch.zeros_like(g) for _ in range(world)]
                dist.all_gather(buf, g.contiguous())
                gathered.append(sum(buf) / world)
            red.reset()
            ((net(x) - y) ** 2).mean().backward()                  # hooks launch the bucket all-reduces
            red.wait()
            for p, want in zip(net.parameters(), gathered):
                assert torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6), (rank, step, (p.grad - want).abs().max())
            ref_grads = [p.grad.clone() for p in net.parameters()]
        # gradient accumulation (the reference's `accumulate` micro-steps): two backward passes, ONE exchange
        xs = [torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(500 + rank + 10 * j)) for j in range(2)]
        ys = [torch.randn(4, 5, generator=torch.Generator().manual_seed(700 + rank + 10 * j)) for j in range(2)]
        local = [torch.autograd.grad(((net(x) - y) ** 2).mean(), list(net.parameters())) for x, y in zip(xs, ys)]
        want = []
        for g0, g1 in zip(*local):
            buf = [torch.zeros_like(g0) for _ in range(world)]
            dist.all_gather(buf, (g0 + g1).contiguous())
            want.append(sum(buf) / world)
        red.reset()
        with red.no_sync():
            ((net(xs[0]) - ys[0]) ** 2).mean().backward()
        ((net(xs[1]) - ys[1]) ** 2).mean().backward()
        red.wait()
        for p, w in zip(net.parameters(), want):
            assert torch.allclose(p.grad, w, rtol=1e-5, atol=1e-6), (rank, "accumulate", (p.grad - w).abs().max())
        # a second backward WITHOUT no_sync / reset must not silently add an un-reduced gradient
        try:
            ((net(xs[0]) - ys[0]) ** 2).mean().backward()
            raise AssertionError("second backward after the exchange did not raise")
        except RuntimeError as e:
            assert "already all-reduced" in str(e)
        red.reset()
        ((net(x) - y) ** 2).mean().backward()
        red.wait()
        ref_grads = [p.grad.clone() for p in net.parameters()]
        # the protocol of a captured-and-replayed backward (train.GraphedTrainStep): the "capture" backward only leaves marks where buckets
        # complete; a "replay" writes gradients with no hooks at all; then exchange(bucket) in mark order, SUM on the wire, the consumer
        # divides (defer_average: optim.FusedSGD.grad_scale = 1 / world)
        red.reset()
        marks = []
        red.begin_marks(marks.append)
        ((net(x) - y) ** 2).mean().backward()                      # "capture": hooks fire, nothing is exchanged
        order, rest = red.end_marks()
        assert order == marks and sorted(order + rest) == list(range(len(red.buckets))) and not rest
        assert order[0] == 0                                       # reverse registration order: the last layer's bucket completes first
        local = torch.autograd.grad(((net(x) - y) ** 2).mean(), list(net.parameters()))
        want = []
        for g in local:
            buf = [torch.zeros_like(g) for _ in range(world)]
            dist.all_gather(buf, g.contiguous())
            want.append(sum(buf) / world)
        red.defer_average = True
        for rep in range(2):                                       # two "replays"
            for p, g in zip(net.parameters(), local):
                p.grad.copy_(g)                                    # what the replayed kernels do: write into the bucket views
            for bi in order:
                red.exchange(bi)
            red.wait_works()
            for p, w in zip(net.parameters(), want):
                assert torch.allclose(p.grad / world, w, rtol=1e-5, atol=1e-6), (rank, "marks", rep, (p.grad / world - w).abs().max())
        red.defer_average = False
        red.reset()
        ((net(x) - y) ** 2).mean().backward()
        red.wait()
        ref_grads = [p.grad.clone() for p in net.parameters()]
        # replicas agree bit for bit after the exchange
        for g in ref_grads:
            buf = [torch.zeros_like(g) for _ in range(world)]
            dist.all_gather(buf, g)
            assert torch.equal(buf[0], buf[1])
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_bucket_plan_for_lead_yolo_s():
    """bucket plan on the real parameter list: reverse order, ~2 MB buckets, small first bucket"""
    import lead_yolo_amd as L
    from lead_yolo_amd.ddp import GradReducer
    m = L.Model(L.load_cfg(scale="s"))
    red = GradReducer(m.parameters())
    plan = red.plan()
    assert sum(b["n_tensors"] for b in plan) == 186
    assert sum(b["bytes"] for b in plan) == 4 * 3135478
    assert plan[0]["bytes"] <= 1 << 20 and len(plan) >= 6
    first = red.buckets[0]["params"][0]
    assert first is list(m.parameters())[-1]                      # Detect's last bias is reduced first
