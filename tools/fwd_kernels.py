"""Device time per kernel of ONE eager eval forward (torch profiler, device activity): the launches a GraphedForward replays.
    python tools/fwd_kernels.py [f32|bf16] [bs] [top] [parts]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
bf = len(sys.argv) > 1 and sys.argv[1] == "bf16"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
parts = int(sys.argv[4]) if len(sys.argv) > 4 else 1      # kernel choice as inside a GraphedForward of `parts` concurrent sub-batches
model = B.build_model("s", dev)
from lead_yolo_amd import ops
ops.CONCURRENT_PARTS = parts
x = B.synth_batch(bs, 640, 0, dev)
if bf:
    x = x.to(torch.bfloat16)
with torch.no_grad():
    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        model(x)
        torch.cuda.synchronize()
from demangle import demangle, short
acc, cnt = collections.Counter(), collections.Counter()
evs = [ev for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA]
names = demangle([ev.name for ev in evs])
for ev, nm in zip(evs, names):
    nm = short(nm) if "ly_" in nm else nm
    acc[nm] += ev.device_time
    cnt[nm] += 1
tot = sum(acc.values())
own = sum(v for k, v in acc.items() if k.startswith("ly_"))
print(f"one eager eval forward, lead-yolo-s bs={bs} 640x640 {'bf16' if bf else 'f32'}: kernels={sum(cnt.values())} "
      f"busy={tot / 1e3:.3f} ms  (ly_* {own / 1e3:.3f} ms, ATen / memcpy {(tot - own) / 1e3:.3f} ms)")
for k, v in acc.most_common(top):
    print(f"{k[:130]:<130} {cnt[k]:4d}  {v / cnt[k]:8.1f} us  {v:9.1f} us")
