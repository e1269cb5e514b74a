"""k = 1 RFCBAMConv statistics + pooling pass (ly_rfcbam_pre1*) over pixels per block, bs = 64 bf16, graph replay"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lead_yolo_amd import ops
dev = torch.device("cuda:0")

def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    for _ in range(3): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for c, hw in ((160, 20), (256, 40)):
    x = torch.randn(64, c, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xr, ld = ops.rows(x)
    a1, b1 = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    for px in (16, 32, 64, 128):
        ops.PRE1_PIXELS = px
        t = timeit(lambda: ops.rfcbam_stats(xr, ld, 64, hw, hw, c, 1, 1, a1=a1, b1=b1, gap=True))
        print(f"C={c} {hw}x{hw}: {px} pixels per block: {t:.1f} us  ({64 * hw * hw * c * 2 / t / 1e6:.2f} TB/s)")
