"""Which python lines of the package issue the small torch launches (fills, adds, copies, casts) of one training step.
    python tools/glue_ops.py [f32|bf16]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import lead_yolo_amd as L
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
amp = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else None
model = B.build_model("s", dev, train=True)
opt = L.smart_optimizer(model, "SGD", 0.01, 0.937, 5e-4)
cl = L.ComputeLoss(model)
imgs = B.synth_u8(16, 640, 0).to(dev)
tg = B.synth_targets(16, 1).to(dev)
for _ in range(3):
    L.train_step(model, cl, opt, imgs, tg, amp=amp)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    L.train_step(model, cl, opt, imgs, tg, amp=amp)
    torch.cuda.synchronize()
acc = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    if not any(k in ev.name for k in ("zero", "fill", "add", "copy", "mul", "cat", "sum", "clone", "contiguous", "to", "div", "sub", "empty_like", "full", "einsum", "permute")):
        continue
    frame = next((f for f in ev.stack if "lead-yolo_amd" in f or "lead_yolo_amd" in f), None)
    if frame is None:
        frame = "<autograd / torch internals>"
    acc[(frame.split("lead-yolo_amd/")[-1][:90], ev.name)] += 1
for (frame, name), n in acc.most_common(60):
    print(f"{n:4d}  {name:<22} {frame}")
