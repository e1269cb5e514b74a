"""RFCBAMConv (k=3) main kernel at the two lead-yolo-s shapes with the ablation switches of ly_debug_set_rf3
(1 skip regenerate, 2 generate weights through LDS, 4 skip staging, 8 two MT=2 groups instead of MT=4).  Module time per call inside a replayed hipGraph (SE + stats + main kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
from lead_yolo_amd import capi
dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def timeit(fn, iters=20, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3
for c1, c2, hw in ((128, 128, 80), (256, 256, 40)):
    m = L.RFCBAMConv(c1, c2, 3, 2).to(dev).eval()
    x = torch.randn(bs, c1, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for dbg in (0,):
            capi.lib().ly_debug_set_rf3(dbg)
            print(f"{c1}->{c2} @{hw} bs={bs} dbg={dbg} (skip gen={dbg&1} ldsw={(dbg>>1)&1} stage={(dbg>>2)&1} mt2x2={(dbg>>3)&1}): module {timeit(lambda: m(x)):8.1f} us")
capi.lib().ly_debug_set_rf3(0)
