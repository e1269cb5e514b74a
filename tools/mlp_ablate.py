import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
from lead_yolo_amd import capi
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for c, hw in ((24, 160), (80, 40)):
    m = L.BasicStage(c, 1).to(dev).eval()
    x = torch.randn(32, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for dbg in (0, 1, 2, 4, 8, 3, 7, 15):
            capi.lib().ly_debug_set_mlp(dbg)
            print(f"C={c} dbg={dbg:2d} (skip pconv={dbg&1} mlp={(dbg>>1)&1} halo={(dbg>>2)&1} store={(dbg>>3)&1}): {timeit(lambda: m(x)):8.1f} us")
capi.lib().ly_debug_set_mlp(0)
