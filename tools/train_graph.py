"""Experiment: forward and backward of the model as two hipGraphs (torch.cuda.make_graphed_callables), loss + optimiser eager."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = L.Model(L.load_cfg(scale="s")).to(dev).train()
opt = L.smart_optimizer(m, "SGD", 0.01, 0.937, 5e-4 * bs / 64)
cl = L.ComputeLoss(m)
g = torch.Generator().manual_seed(0)
imgs = (torch.randint(0, 256, (bs, 3, 640, 640), dtype=torch.uint8, generator=g).float() / 255).to(dev)
nb = 7 * bs
tg = torch.cat((torch.sort(torch.randint(0, bs, (nb, 1), generator=g).float(), 0)[0], torch.zeros(nb, 1),
                torch.rand(nb, 2, generator=g) * 0.8 + 0.1, torch.rand(nb, 2, generator=g) * 0.2 + 0.02), 1).to(dev)
def step(model):
    pred = model(imgs)
    loss, _ = cl(pred, tg)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 10.0)
    opt.step(); opt.zero_grad(set_to_none=True)
    return loss
def timeit(model, n=10):
    for _ in range(3): step(model)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): l = step(model)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, float(l)
print("eager   : %.2f ms/step (loss %.4f)" % timeit(m))
gm = torch.cuda.make_graphed_callables(m, (imgs,), num_warmup_iters=3)
print("graphed : %.2f ms/step (loss %.4f)" % timeit(gm))
