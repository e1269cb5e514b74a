"""Times the full training step (forward, loss, backward, clip, SGD) of lead-yolo-s on synthetic COCO-shaped data."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L

ap = argparse.ArgumentParser()
ap.add_argument("--bs", type=int, default=32)
ap.add_argument("--size", type=int, default=640)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--scale", default="s")
ap.add_argument("--fwd-only", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = L.Model(L.load_cfg(scale=a.scale)).to(dev).train()
opt = L.smart_optimizer(m, "SGD", 0.01, 0.937, 5e-4 * a.bs / 64)
cl = L.ComputeLoss(m)
g = torch.Generator().manual_seed(0)
imgs = torch.randint(0, 256, (a.bs, 3, a.size, a.size), dtype=torch.uint8, generator=g).to(dev)
nb = 7 * a.bs
tg = torch.cat((torch.sort(torch.randint(0, a.bs, (nb, 1), generator=g).float(), 0)[0], torch.zeros(nb, 1),
                torch.rand(nb, 2, generator=g) * 0.8 + 0.1, torch.rand(nb, 2, generator=g) * 0.2 + 0.02), 1).to(dev)


def step():
    if a.fwd_only:
        with torch.no_grad():
            return m(imgs.float() / 255)
    return L.train_step(m, cl, opt, imgs, tg)


for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"bs={a.bs} size={a.size} {'train-fwd' if a.fwd_only else 'train-step'}: {dt*1e3:.2f} ms/step  {a.bs/dt:.1f} img/s  "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
