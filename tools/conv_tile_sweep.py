"""conv3x3 forward time by patch shape (ops.pick_conv_tile's choice vs alternatives), ten launches in one hipGraph.  Dev tool, GPU box.
    python tools/conv_tile_sweep.py [batch=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
from lead_yolo_amd import ops, pack

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
DT = torch.bfloat16
REP = 10


def graph_time(fn, iters=5):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(REP):
            fn()
    for _ in range(2):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / REP * 1e3


def case(hw, c, n, tiles):
    M = B * hw * hw
    x = torch.randn(M, c, device=dev).to(DT)
    w = torch.randn(n, c, 3, 3, device=dev) / (9 * c) ** 0.5
    wp = pack.frag_pack3(pack.conv_taps_matrix(w, 32), planes=1)
    out = torch.empty(M, n, device=dev, dtype=DT)
    sc = torch.ones(n, device=dev)
    sh = torch.zeros(n, device=dev)
    fn = lambda: ops.conv3x3(M=M, H=hw, W=hw, Cin=c, N=n, x=x, ldx=c, wp=wp, out=out, ldo=n, e_scale=sc, e_shift=sh, act=2)
    ops._CONV_TILE_CACHE.pop((hw, hw), None)
    base = ops.pick_conv_tile(hw, hw)
    ref = None
    for t in [base] + [t for t in tiles if t != base]:
        ops._CONV_TILE_CACHE[(hw, hw)] = t
        fn()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        ok = torch.equal(ref, out)
        us = graph_time(fn)
        blocks = -(-hw // t[0]) * -(-hw // t[1])
        print(f"{hw}x{hw} C={c} N={n} tile {t[0]:2d}x{t[1]:2d} ({t[0] * t[1]:3d} px, {blocks:3d} blocks/img, {blocks * -(-(t[0] * t[1]) // 16):3d} mfma tiles/img) {us:7.1f} us  same bits {ok}"
              + ("   <- pick_conv_tile" if t == base else ""), flush=True)
    ops._CONV_TILE_CACHE.pop((hw, hw), None)


if len(sys.argv) > 2 and sys.argv[2] == "shapes80":
    case(40, 128, 128, [(10, 8), (8, 10), (4, 20), (5, 16), (16, 5), (20, 4), (2, 40), (40, 2)])
    case(20, 256, 256, [(20, 4), (4, 20), (8, 10), (10, 8), (5, 16), (16, 5)])
    case(40, 64, 64, [(10, 8), (8, 10), (4, 20), (20, 4)])
else:
    case(40, 128, 128, [(10, 8), (20, 6), (10, 10), (8, 14), (14, 8), (20, 5), (8, 10), (10, 12), (5, 8), (8, 16), (16, 8), (13, 8)])
    case(20, 256, 256, [(20, 4), (20, 6), (10, 12), (20, 5), (10, 10), (7, 10), (10, 8), (5, 20)])
    case(80, 64, 64, [(16, 8), (8, 16), (10, 8), (10, 12), (20, 6), (12, 10)])
