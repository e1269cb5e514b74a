"""Who issues the fills / copies of a training step: for every aten::fill_ / aten::copy_ event of one step, the chain of enclosing profiler
ranges (autograd node or python-level op).    python tools/glue_parents.py [f32|bf16] [fill_|copy_|add|...]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import lead_yolo_amd as L
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
amp = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else None
what = sys.argv[2] if len(sys.argv) > 2 else "fill_"
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
    if ev.name != "aten::" + what:
        continue
    chain, p = [], ev.cpu_parent
    while p is not None:
        chain.append(p.name.replace("autograd::engine::evaluate_function: ", "bwd:"))
        p = p.cpu_parent
    acc[(" <- ".join(chain[:4])[:150], str(ev.input_shapes)[:60])] += 1
for (chain, shp), n in acc.most_common(50):
    print(f"{n:4d}  {shp:<60} {chain}")
