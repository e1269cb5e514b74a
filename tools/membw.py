import torch, time
dev = "cuda"
for mb in (52, 210, 840):
    n = mb * 1024 * 1024 // 4
    a = torch.randn(n, device=dev); b = torch.empty_like(a)
    for name, fn, bytes_ in (("copy", lambda: b.copy_(a), 2 * n * 4), ("sum", lambda: a.sum(), n * 4), ("mul_", lambda: a.mul_(1.0001), 2 * n * 4),
                             ("fill", lambda: b.fill_(1.0), n * 4)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{mb:4d} MB {name:5s} {us:8.1f} us  {bytes_ / us / 1e6:8.1f} GB/s")
