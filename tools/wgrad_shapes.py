"""Per-shape time of every ly_wgrad call of one training step (lead-yolo-s, HIP events around each C-ABI call)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
from lead_yolo_amd import ops
dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
amp = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else None
from lead_yolo_amd import capi
m = L.Model(L.load_cfg(scale="s")).to(dev).train()
opt = L.smart_optimizer(m, "SGD", 0.01, 0.937, 5e-4 * bs / 64)
cl = L.ComputeLoss(m)
g = torch.Generator().manual_seed(0)
imgs = torch.randint(0, 256, (bs, 3, 640, 640), dtype=torch.uint8, generator=g).to(dev)
nb = 7 * bs
tg = torch.cat((torch.sort(torch.randint(0, bs, (nb, 1), generator=g).float(), 0)[0], torch.zeros(nb, 1),
                torch.rand(nb, 2, generator=g) * 0.8 + 0.1, torch.rand(nb, 2, generator=g) * 0.2 + 0.02), 1).to(dev)
orig = ops.wgrad
def named(**k):
    ops._WG_NAME = f"wgrad N={k['N']:3d} K={k.get('ks', 1) ** 2 * k['Cin']:4d} M={k['M']:7d} ks={k.get('ks', 1)} s={k.get('stride', 1)} lddu={k['lddu']} ldx={k['ldx']}" \
                   f"{' nchw' if k.get('nchw') else ''}{' up2' if k.get('up2') else ''}"
    return orig(**k)
ops.wgrad = named
import lead_yolo_amd.grad as G
G.ops.wgrad = named
T = ops._Timed
class T2(T):
    def __init__(self, name, flops, nbytes):
        super().__init__(getattr(ops, "_WG_NAME", name) + " " + name.split("<")[1][:-1] if name.startswith("ly_wgrad") else name, flops, nbytes)
ops._Timed = T2
for _ in range(2):
    L.train_step(m, cl, opt, imgs, tg, amp=amp)
torch.cuda.synchronize()
ops.PROFILE = []
L.train_step(m, cl, opt, imgs, tg, amp=amp)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for name, fl, by, e0, e1 in ops.PROFILE:
    if name.startswith("wgrad"):
        a = agg[name]; a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3; a[2] = by
ops.PROFILE = None
tot = 0.0
for name, (n, us, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += us
    print(f"{name:100s} x{n}  {us / n:8.1f} us  {by / (us / n) / 1e3:7.1f} GB/s")
print(f"total wgrad {tot / 1e3:.2f} ms")
