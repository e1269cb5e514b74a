"""1x1 GEMM forward time by shape, ten launches in one hipGraph (no replay floor).  Dev tool, GPU box.
    python tools/gemm_time.py [bf16|f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lead_yolo_amd import ops, pack

dev = torch.device("cuda:0")
DT = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.bfloat16
PL = 1 if DT == torch.bfloat16 else 2
ES = 2 if DT == torch.bfloat16 else 4
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


def case(hw, b, k, n, stats=False):
    M = b * hw * hw
    a = torch.randn(M, k, device=dev).to(DT)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    wp = pack.frag_pack3(w, planes=PL)
    out = torch.empty(M, n, device=dev, dtype=DT)
    sc = torch.ones(n, device=dev)
    sh = torch.zeros(n, device=dev)
    st = ops.new_stats(n, dev) if stats else None
    fn = lambda: ops.gemm(M=M, H=hw, W=hw, K=k, N=n, a0=a, lda0=k, k0=k, wp=wp, out=out, ldo=n, e_scale=sc, e_shift=sh, act=0 if stats else 2, stats=st)
    us = graph_time(fn)
    print(f"M={M:7d} K={k:4d} N={n:4d} {'stats' if stats else '     '} {us:7.1f} us  {ES * M * (k + n) / us / 1e3:7.0f} GB/s  {2.0 * M * k * n / us / 1e6:7.1f} TFLOP/s", flush=True)


for st in (False, True):
    case(20, 64, 512, 512, st)
    case(20, 64, 256, 256, st)
    case(20, 64, 320, 160, st)
    case(20, 64, 160, 320, st)
    case(40, 64, 256, 256, st)
    case(40, 64, 128, 128, st)
    case(80, 64, 128, 128, st)
    case(80, 64, 64, 64, st)
