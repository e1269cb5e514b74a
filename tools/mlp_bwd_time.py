"""Fused vs unfused MLPBlock backward (csrc/ly_mlpblock_bwd.hpp), bf16, the BasicStage shapes of lead-yolo-s at bs = 64: per-kernel HIP-event
times of one forward + backward of the module:  python tools/mlp_bwd_time.py [C ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L  # noqa: E402
from lead_yolo_amd import ops  # noqa: E402

SHAPES = {24: (64, 160, 160), 40: (64, 80, 80), 80: (64, 40, 40), 160: (64, 20, 20)}


def run(c, fused, iters=5):
    ops.MLP_BWD_FUSED = fused
    n, h, w = SHAPES[c]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = L.BasicStage(c, 1).to(dev).train()
    opt = L.smart_optimizer(m, "SGD", 0.01, 0.937, 5e-4)          # installs the gradient sink (parameter gradients written in place)
    x = torch.randn(n, c, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn(n, c, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        y.backward(r)
        x.grad = None
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
        with torch.cuda.graph(g, stream=s):
            step()
    torch.cuda.synchronize()
    g.replay()
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    total = e0.elapsed_time(e1) / iters * 1e3
    ops.PROFILE = []
    step()
    torch.cuda.synchronize()
    recs, ops.PROFILE = ops.PROFILE, None
    print(f"C={c} {n}x{h}x{w} fused={fused}: forward + backward {total:8.1f} us (graph replay)")
    for name, _, nbytes, a, b, _ in recs:
        t = a.elapsed_time(b) * 1e3
        print(f"      {t:8.1f} us  {nbytes / t / 1e6:8.2f} TB/s  {name}")
    del opt


if __name__ == "__main__":
    cs = [int(a) for a in sys.argv[1:]] or [24, 40]
    for c in cs:
        for fused in (False, True):
            run(c, fused)
