"""BatchNorm / activation backward passes alone (ly_bnact_bwd_reduce / _apply, ly_bnact_fwd) on the step's shapes, from HBM (buffers rotated so that
nothing is resident in the 256 MiB Infinity Cache) and back to back inside a hipGraph (no launch gaps): python tools/bnact_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lead_yolo_amd import ops                               # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def graph_time(fn, reps=10, iters=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st), torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / reps * 1e3


for rows, c in [(409600, 128), (409600, 64), (102400, 256), (102400, 128), (25600, 512), (1638400, 48)]:
    nb = max(2, int(600e6 // (rows * c * 2 * 2)) + 1)           # > 256 MiB in rotation
    us = [torch.randn(rows, c, device=dev).to(BF) for _ in range(nb)]
    dys = [torch.randn(rows, c, device=dev).to(BF) for _ in range(nb)]
    dus = [torch.empty(rows, c, device=dev, dtype=BF) for _ in range(2)]
    a, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    k = [0]

    def nxt():
        k[0] = (k[0] + 1) % nb
        return us[k[0]], dys[k[0]]
    out = []
    for act in (2, 1, 0):
        def red():
            u, dy = nxt()
            ops.bnact_bwd_reduce(dy, c, u, c, rows, c, a, b, act)

        def app():
            u, dy = nxt()
            ops.bnact_bwd_apply(dy, c, u, c, rows, c, a, b, act, a, b, a, dus[0], c)

        def fwd():
            u, _ = nxt()
            ops.bnact_fwd(u, c, rows, c, a, b, act, dus[1], c)
        tr, ta, tf = graph_time(red), graph_time(app), graph_time(fwd)
        mb = rows * c * 2 / 1e6
        out.append(f"act {act}: reduce {tr:.1f} us ({2 * mb / tr:.0f} GB/s) apply {ta:.1f} us ({3 * mb / ta:.0f}) fwd {tf:.1f} us ({2 * mb / tf:.0f})")
    cp = graph_time(lambda: dus[0].copy_(nxt()[0]))
    sm = graph_time(lambda: nxt()[0].float().sum())
    print(f"rows={rows} C={c} ({rows * c * 2 / 1e6:.0f} MB per tensor): " + " | ".join(out) + f" | torch copy {cp:.1f} us ({2 * rows * c * 2 / 1e6 / cp:.0f} GB/s)", flush=True)
