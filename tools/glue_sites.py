"""Which python lines of the package call the small torch ops (fills, copies, casts) of one training step: wraps the python entry points
and counts call sites.    python tools/glue_sites.py [f32|bf16]"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import lead_yolo_amd as L
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
acc = collections.Counter()
ON = [False]


def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "lead-yolo_amd" in f.filename or "lead_yolo_amd" in f.filename:
            return f"{os.path.basename(f.filename)}:{f.lineno} {f.line[:70]}"
    return "<other>"


def wrap(owner, name):
    orig = getattr(owner, name)

    def w(*a, **k):
        if ON[0]:
            ON[0] = False
            acc[(name, site())] += 1
            ON[0] = True
        return orig(*a, **k)
    setattr(owner, name, w)


for n in ("zeros", "zeros_like", "ones", "full", "empty_like", "cat", "stack", "where"):
    wrap(torch, n)
for n in ("zero_", "new_zeros", "fill_", "copy_", "clone", "contiguous", "to", "float", "sum", "mean", "add_", "mul_", "__add__", "__mul__", "__sub__",
          "__truediv__", "type"):
    wrap(torch.Tensor, n)
ON[0] = True
L.train_step(model, cl, opt, imgs, tg, amp=amp)
ON[0] = False
torch.cuda.synchronize()
for (name, s), n in acc.most_common(70):
    print(f"{n:4d}  {name:<12} {s}")
