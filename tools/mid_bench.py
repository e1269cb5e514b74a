"""Timing of the small steps between the RFCBAMConv statistics pass and its contraction (ly_rfcbam_mid = SE linears + get_weight map; ly_rfa_map
alone): python tools/mid_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lead_yolo_amd import ops                               # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=200):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for n, c, hk, wk, slices, hw in [(64, 128, 120, 120, 50, 6400), (64, 256, 60, 60, 15, 1600), (64, 160, 20, 20, 8, 400), (64, 256, 40, 40, 8, 1600), (1, 128, 120, 120, 50, 6400)]:
    r = max(c // 16, 1)
    part = torch.randn(n, slices, c, device=dev)
    wa, wb = torch.randn(r, c, device=dev), torch.randn(c, r, device=dev)
    mm = torch.rand(n, hk, wk, 2, device=dev)
    w18 = torch.randn(18, device=dev)
    g = torch.cuda.CUDAGraph()
    ops.rfcbam_mid(part, hw, wa, wb, r, mm, w18)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st), torch.cuda.graph(g):
        for _ in range(10):
            ops.rfcbam_mid(part, hw, wa, wb, r, mm, w18)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st), torch.cuda.graph(g2):
        for _ in range(10):
            ops.rfa_map(mm, w18)
    print(f"n={n} C={c} map {hk}x{wk} slices={slices}: mid {timeit(g.replay, 50) / 10:.1f} us | rfa_map alone {timeit(g2.replay, 50) / 10:.1f} us (10 back to back in a graph)")
