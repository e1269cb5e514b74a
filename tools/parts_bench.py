"""Serving-mode forward: how many concurrent sub-batches (graph.GraphedForward parts)?  python tools/parts_bench.py [bf16|f32] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L                                   # noqa: E402
from oracle import synth                                    # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.manual_seed(0)
m = L.Model(L.load_cfg(scale="s")).to(dev).eval()
x = (synth.synth_images(bs, 640, 5).float() / 255).to(dev).to(dt)
for parts in (1, 2, 4, 8, 16):
    if bs % parts or bs // parts < 1:
        continue
    g = L.GraphedForward(m, x, parts=parts)
    for _ in range(3):
        g()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(30):
        g()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    print(f"{dt} bs={bs} parts={parts}: {ms:.4f} ms  {bs / ms * 1e3:.0f} img/s", flush=True)
    del g
