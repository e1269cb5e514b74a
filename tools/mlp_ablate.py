"""MLPBlock kernel at the four lead-yolo-s shapes: pixel-tile variants (ly_debug_set_mlp_tile) and the ablation switches
(ly_debug_set_mlp: 4 skip halo staging, 8 skip stores).  Times are per launch inside a replayed hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
from lead_yolo_amd import capi
dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def timeit(fn, iters=20, reps=10):
    """kernel time: `reps` calls captured in one hipGraph, replayed `iters` times (no host gaps between launches)"""
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
lib = capi.lib()
for c, hw in ((24, 160), (40, 80), (80, 40), (160, 20)):
    m = L.BasicStage(c, 1).to(dev).eval()
    x = torch.randn(bs, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = None
        for tile in (0, 8, 2, 4):
            lib.ly_debug_set_mlp_tile(tile)
            y = m(x)
            if ref is None: ref = y
            same = bool((y == ref).all())
            row = [f"tile={tile}", f"bitwise={same}"]
            for dbg in (0, 4, 8):
                lib.ly_debug_set_mlp(dbg)
                row.append(f"dbg{dbg}: {timeit(lambda: m(x)):7.1f} us")
            lib.ly_debug_set_mlp(0)
            print(f"C={c:3d} {hw}x{hw} bs={bs}  " + "  ".join(row))
lib.ly_debug_set_mlp_tile(0)
