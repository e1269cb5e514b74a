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
from torch.profiler import profile, ProfilerActivity

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
step = L.GraphedTrainStep(model, cl, opt, imgs, tg, ema=ema, amp=torch.bfloat16, warmup=2, reducer=red, world_size=1)
print(f"buckets {len(red.buckets)}: {len(step._marked)} released from events inside graph A (order {step._marked}), {len(step._unmarked)} after it")
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
evs.sort(key=lambda e: e.time_range.start)
t0 = evs[0].time_range.start
ly = [e for e in evs if "ly_" in e.name]
if not any("ly_optim" in e.name for e in ly):
    import collections
    print("no ly_optim kernel in the device trace; names seen:", collections.Counter(e.name[:50] for e in evs).most_common(12))
    sys.exit(0)
opt_start = next(e.time_range.start for e in ly if "ly_optim" in e.name)
a_end = max(e.time_range.end for e in ly if e.time_range.start < opt_start)
print(f"graph A kernels span 0 .. {(a_end - t0) / 1e3:.3f} ms; optimiser starts at {(opt_start - t0) / 1e3:.3f} ms; {len(evs)} device events")
n = 0
for e in evs:
    nm = e.name
    if "ccl" in nm.lower() or "allreduce" in nm.lower() or "AllReduce" in nm:
        n += 1
        s = (e.time_range.start - t0) / 1e3
        print(f"  exchange kernel {n}: starts at {s:.3f} ms ({'INSIDE' if e.time_range.start < a_end else 'after'} graph A), {e.device_time:.1f} us   {nm[:70]}")
if not n:
    print("  (no RCCL kernel appeared in the device trace: a one-rank all-reduce is elided by the library; the host-side release path still ran)")
dist.destroy_process_group()
