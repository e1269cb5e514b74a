import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
dev = torch.device("cuda:0")
bs = 64
m = L.Model(L.load_cfg(scale="s")).to(dev).train()
cl = L.ComputeLoss(m)
g = torch.Generator().manual_seed(0)
nb = 7 * bs
tg = torch.cat((torch.sort(torch.randint(0, bs, (nb, 1), generator=g).float(), 0)[0], torch.zeros(nb, 1),
                torch.rand(nb, 2, generator=g) * 0.8 + 0.1, torch.rand(nb, 2, generator=g) * 0.2 + 0.02), 1).to(dev)
pred = [torch.randn(bs, 3, s, s, 6, device=dev, requires_grad=True) for s in (80, 40, 20)]
for _ in range(3):
    loss, _ = cl(pred, tg); loss.backward()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    loss, _ = cl(pred, tg)
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(20):
    loss, _ = cl(pred, tg); loss.backward()
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"loss forward {(t1-t0)/20*1e3:.2f} ms, forward+backward {(t2-t1)/20*1e3:.2f} ms (bs={bs}, {nb} targets)")
