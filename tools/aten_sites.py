"""Which lines of this package issue the ATen launches (copies, fills, small elementwise kernels) of ONE steady-state eager training step:
torch profiler with Python stacks, device time of every aten:: op grouped by the innermost lead-yolo_amd frame.
    python tools/aten_sites.py [f32|bf16] [bs] [top]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import lead_yolo_amd as L
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
amp = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else None
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
model = B.build_model("s", dev, train=True)
opt = L.smart_optimizer(model, "SGD", 0.01, 0.937, 5e-4, fused=True)
cl = L.ComputeLoss(model)
ema = L.ModelEMA(model)
imgs = B.synth_u8(bs, 640, 0).to(dev)
tg = B.synth_targets(bs, 1).to(dev)
for _ in range(3):
    L.train_step(model, cl, opt, imgs, tg, ema=ema, amp=amp)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    L.train_step(model, cl, opt, imgs, tg, ema=ema, amp=amp)
    torch.cuda.synchronize()
acc, cnt = collections.Counter(), collections.Counter()
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=24):
    if not ev.key.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    site = next((s for s in ev.stack if "lead-yolo_amd" in s), ev.stack[0] if ev.stack else "?")
    site = site.split("lead-yolo_amd/")[-1]
    if site == "?":
        site = str(ev.input_shapes)                 # (autograd-engine thread: no Python frames) the operand shapes identify the op
    acc[(site, ev.key)] += ev.self_device_time_total
    cnt[(site, ev.key)] += ev.count
tot = sum(acc.values())
print(f"ATen launches of one eager step (lead-yolo-s bs={bs} 640x640 {'bf16' if amp else 'f32'}): {sum(cnt.values())} ops, {tot / 1e3:.3f} ms device time")
for k, v in acc.most_common(top):
    print(f"{k[0][:90]:<90} {k[1]:<28} {cnt[k]:4d}  {v:9.1f} us")
