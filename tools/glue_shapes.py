"""Shapes of the small torch ops (fills, copies, adds) of one training step: python tools/glue_shapes.py [f32|bf16]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import lead_yolo_amd as L
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
amp = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else None
model = B.build_model("s", dev, train=True)
opt = L.smart_optimizer(model, "SGD", 0.01, 0.937, 5e-4, fused=True)
cl = L.ComputeLoss(model)
imgs = B.synth_u8(16, 640, 0).to(dev)
tg = B.synth_targets(16, 1).to(dev)
for _ in range(3):
    L.train_step(model, cl, opt, imgs, tg, amp=amp)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    L.train_step(model, cl, opt, imgs, tg, amp=amp)
    torch.cuda.synchronize()
acc = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::sum", "aten::cat", "aten::clone", "aten::index_put_"):
        acc[(ev.name, str(ev.input_shapes)[:110])] += 1
for (name, shp), n in acc.most_common(80):
    print(f"{n:4d}  {name:<14} {shp}")
