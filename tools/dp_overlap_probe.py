"""Where do the gradient exchanges of the captured data-parallel step start?  One rank (an RCCL group of world size 1 on this GPU: every
bucket's all-reduce is really launched), lead-yolo-s bs=64 bf16, train.GraphedTrainStep with a ddp.GradReducer: graph A (forward + backward
with bucket-completion event nodes) -> per-bucket all-reduce released from those events on the communication stream -> graph B (optimiser).
Prints, from one profiled step (torch profiler, device activity), the start of every RCCL kernel relative to the span of graph A's kernels.
    python tools/dp_overlap_probe.py [bs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench as B
import lead_yolo_amd as L

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29871", rank=0, world_size=1, device_id=dev)
model = B.build_model("s", dev, train=True)
opt = L.smart_optimizer(model, "SGD", 0.01, 0.937, 5e-4)
cl = L.ComputeLoss(model)
ema = L.ModelEMA(model)
red = L.GradReducer(list(model.parameters())).attach()
red.exchange_single = True
imgs = B.synth_u8(bs, 640, 0).to(dev)
tg = B.synth_targets(bs, 1).to(dev)
step = L.GraphedTrainStep(model, cl, opt, imgs, tg, ema=ema, amp=torch.bfloat16, warmup=2, reducer=red, world_size=1, dp_exchange="overlapped")    # this tool looks at the overlapped form
print(f"buckets {len(red.buckets)}: {len(step._marked)} released from events inside graph A (order {step._marked}), {len(step._unmarked)} after it")
for _ in range(3):
    step()
torch.cuda.synchronize()
# HIP events: start of graph A, end of graph A (before the exchange wait), release time of every bucket on the communication stream
rows = []
for _ in range(5):
    step.probe = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    # (the step's own code path, with an end-of-graph-A event in between)
    step._load(None, None)
    step.optimizer._sync_hyper()
    step.graph.replay()
    e1.record()
    step._micro += 1
    from lead_yolo_amd import capi, pack
    lib, cur, comm = capi.lib(), torch.cuda.current_stream(), step._comm
    with torch.cuda.stream(comm):
        for bi in step._marked:
            capi.check(lib.ly_stream_wait_event(capi._P(comm.cuda_stream), step._events[bi]), "wait")
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(comm)
            step.probe.append((bi, ev))
            red.exchange(bi)
    red.wait_works()
    cur.wait_stream(comm)
    step.opt_graph.replay()
    pack.touch_weights()
    step._micro = 0
    e2 = torch.cuda.Event(enable_timing=True)
    e2.record()
    torch.cuda.synchronize()
    rows.append((e0.elapsed_time(e1), e0.elapsed_time(e2), [(bi, e0.elapsed_time(ev)) for bi, ev in step.probe]))
a_ms, all_ms, rel = rows[-1]
print(f"graph A (forward + backward) runs 0 .. {a_ms:.3f} ms; the step (A, exchange, optimiser graph B) ends at {all_ms:.3f} ms")
for bi, t in rel:
    nb = red.buckets[bi]["flat"].numel() * 4
    print(f"  bucket {bi} ({nb / 1e6:.2f} MB, {len(red.buckets[bi]['params'])} tensors): released + all-reduce queued at {t:.3f} ms = {100 * t / a_ms:.0f} % of graph A"
          f" -> {a_ms - t:.3f} ms of backward still to run")
inside = sum(1 for _, t in rel if t < a_ms)
print(f"{inside} of {len(rel)} buckets released while graph A was still executing (one rank: the RCCL calls are issued, the wire is trivial)")
dist.destroy_process_group()
