"""Which python lines issue the small torch kernels (fill / copy / add ...) of a training step: torch.profiler with stacks,
aten ops counted per calling line of lead_yolo_amd/*.py."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
bs = 8
m = L.Model(L.load_cfg(scale="s")).to(dev).train()
opt = L.smart_optimizer(m, "SGD", 0.01, 0.937, 5e-4 * bs / 64)
cl = L.ComputeLoss(m)
g = torch.Generator().manual_seed(0)
imgs = torch.randint(0, 256, (bs, 3, 256, 256), dtype=torch.uint8, generator=g).to(dev)
nb = 7 * bs
tg = torch.cat((torch.sort(torch.randint(0, bs, (nb, 1), generator=g).float(), 0)[0], torch.zeros(nb, 1),
                torch.rand(nb, 2, generator=g) * 0.8 + 0.1, torch.rand(nb, 2, generator=g) * 0.2 + 0.02), 1).to(dev)
for _ in range(3):
    L.train_step(m, cl, opt, imgs, tg)
torch.cuda.synchronize()

# torch.profiler records no python stacks on this build, and autograd runs the backward on its own thread, so the callers
# are found by wrapping the python entry points (process-wide) and walking the stack to the first frame inside the package
import traceback, functools
cnt = collections.Counter()
def caller():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "lead-yolo_amd/" in fr.filename or "lead_yolo_amd/" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:70]}"
    return "(outside the package)"
def wrap(owner, name, pred=None):
    orig = getattr(owner, name)
    @functools.wraps(orig)
    def f(*a, **k):
        if pred is None or pred(*a, **k):
            cnt[(f"{getattr(owner, '__name__', owner)}.{name}", caller())] += 1
        return orig(*a, **k)
    setattr(owner, name, f)
for n_ in ("zeros", "zeros_like", "full", "full_like", "cat", "ones", "ones_like", "tensor", "stack"):
    wrap(torch, n_)
for n_ in ("zero_", "fill_", "clone", "copy_", "new_zeros", "float"):
    wrap(torch.Tensor, n_, (lambda t, *a, **k: t.dtype != torch.float32) if n_ == "float" else None)
wrap(torch.Tensor, "contiguous", lambda t, *a, **k: not t.is_contiguous(**k))
L.train_step(m, cl, opt, imgs, tg)
torch.cuda.synchronize()
tot = collections.Counter()
for (name, line), c in cnt.items():
    tot[name] += c
print(dict(tot))
for (name, line), c in cnt.most_common(50):
    print(f"{c:5d}  {name:22s} {line}")
