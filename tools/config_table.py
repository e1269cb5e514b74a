"""Throughput of the BASELINE.json configurations this build can run on one MI355X (fp32 I/O, bf16x3 products)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
dev = torch.device("cuda:0")
def fwd(scale, bs, size, graph):
    torch.manual_seed(0)
    m = L.Model(L.load_cfg(scale=scale)).to(dev).eval()
    x = torch.rand(bs, 3, size, size, device=dev)
    step = L.GraphedForward(m, x) if graph else (lambda: m(x))
    with torch.no_grad():
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20 if bs * size * size <= 32 * 640 * 640 else 8
        for _ in range(n): step()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
def train(scale, bs, size):
    torch.manual_seed(0)
    m = L.Model(L.load_cfg(scale=scale)).to(dev).train()
    opt = L.smart_optimizer(m, "SGD", 0.01, 0.937, 5e-4 * bs / 64)
    cl = L.ComputeLoss(m)
    g = torch.Generator().manual_seed(0)
    imgs = torch.randint(0, 256, (bs, 3, size, size), dtype=torch.uint8, generator=g).to(dev)
    nb = 7 * bs
    tg = torch.cat((torch.sort(torch.randint(0, bs, (nb, 1), generator=g).float(), 0)[0], torch.zeros(nb, 1),
                    torch.rand(nb, 2, generator=g) * 0.8 + 0.1, torch.rand(nb, 2, generator=g) * 0.2 + 0.02), 1).to(dev)
    for _ in range(2): L.train_step(m, cl, opt, imgs, tg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): L.train_step(m, cl, opt, imgs, tg)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, torch.cuda.max_memory_allocated() / 2 ** 30
for scale, bs, size in (("n", 1, 640), ("s", 1, 640), ("s", 8, 640), ("s", 32, 640), ("s", 64, 640), ("s", 128, 640), ("l", 16, 1280)):
    e = fwd(scale, bs, size, False); g = fwd(scale, bs, size, True)
    print(f"eval forward  lead-yolo-{scale} bs={bs:3d} {size}: eager {e*1e3:8.2f} ms {bs/e:9.0f} img/s | hipGraph {g*1e3:8.2f} ms {bs/g:9.0f} img/s", flush=True)
for scale, bs, size in (("s", 32, 640), ("s", 64, 640), ("l", 16, 1280)):
    torch.cuda.reset_peak_memory_stats()
    t, mem = train(scale, bs, size)
    print(f"train step    lead-yolo-{scale} bs={bs:3d} {size}: {t*1e3:8.2f} ms {bs/t:9.0f} img/s  peak {mem:.1f} GiB", flush=True)
