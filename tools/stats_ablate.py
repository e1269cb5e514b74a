import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
from lead_yolo_amd import capi, ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (c, hw) in ((128, 80), (256, 40)):
    m = L.RFCBAMConv(c, c, 3, 2).to(dev).eval()
    x = torch.randn(32, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    P = m._packed()
    th, tw = ops.pick_tile(hw // 2, hw // 2)
    print("tile", th, tw)
    for dbg in (0, 1, 2, 3):
        capi.lib().ly_debug_set_stats3(dbg)
        print(f"C={c} dbg={dbg}: stats3 {timeit(lambda: ops.rfcbam_stats(x, c, 32, hw, hw, c, 3, 2, wg=P['wq_stats'], th=th, tw=tw)):8.1f} us")
capi.lib().ly_debug_set_stats3(0)
