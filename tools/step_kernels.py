"""Device time per kernel of ONE steady-state eager training step (torch profiler, device activity), so that model construction /
first-step work does not leak into the per-step table the way it does in a whole-process rocprofv3 trace.
    python tools/step_kernels.py [f32|bf16] [bs] [top] [scale=s] [size=640] [seq]
`seq`: the launches in stream order instead (name, duration), to see which small dependent launches sit next to each other"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import lead_yolo_amd as L
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
amp = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else None
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
top = int(sys.argv[3]) if len(sys.argv) > 3 else 80
scale = sys.argv[4] if len(sys.argv) > 4 else "s"
size = int(sys.argv[5]) if len(sys.argv) > 5 else 640
model = B.build_model(scale, dev, train=True)
opt = L.smart_optimizer(model, "SGD", 0.01, 0.937, 5e-4, fused=True)
cl = L.ComputeLoss(model)
imgs = B.synth_u8(bs, size, 0).to(dev)
tg = B.synth_targets(bs, 1).to(dev)
for _ in range(3):
    L.train_step(model, cl, opt, imgs, tg, amp=amp)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    L.train_step(model, cl, opt, imgs, tg, amp=amp)
    torch.cuda.synchronize()
from demangle import demangle, short
acc, cnt = collections.Counter(), collections.Counter()
evs = [ev for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA and not ev.name.startswith("Optimizer.")]
names = demangle([ev.name for ev in evs])
for ev, nm in zip(evs, names):
    nm = short(nm) if "ly_" in nm else nm
    acc[nm] += ev.device_time
    cnt[nm] += 1
if "seq" in sys.argv:
    order = sorted(zip(evs, names), key=lambda p: p[0].time_range.start)
    for i, (ev, nm) in enumerate(order):
        print(f"{i:4d} {ev.device_time:8.1f} us  {(short(nm) if 'ly_' in nm else nm)[:150]}")
    sys.exit(0)
tot = sum(acc.values())
own = sum(v for k, v in acc.items() if k.startswith("ly_"))
print(f"one steady-state eager optimisation step, lead-yolo-{scale} bs={bs} {size}x{size} {'bf16' if amp else 'f32'}: kernels={sum(cnt.values())} "
      f"busy={tot / 1e3:.3f} ms  (ly_* {own / 1e3:.3f} ms, ATen / memcpy {(tot - own) / 1e3:.3f} ms)")
for k, v in acc.most_common(top):
    print(f"{k[:130]:<130} {cnt[k]:4d}  {v / cnt[k]:8.1f} us  {v:9.1f} us")
