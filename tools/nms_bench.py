"""Timing of the device NMS on val.py-shaped input: python tools/nms_bench.py [bs=32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lead_yolo_amd as L
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_nms import _random_pred
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pred = torch.from_numpy(_random_pred(bs, 25200, 1, 3)).cuda()
for conf in (0.25, 0.001):
    for _ in range(2):
        L.nms_padded(pred, conf, 0.45 if conf > 0.01 else 0.6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        d, c, k = L.nms_padded(pred, conf, 0.45 if conf > 0.01 else 0.6)
    torch.cuda.synchronize()
    print(f"bs={bs} N=25200 conf={conf}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per batch, kept {c.float().mean().item():.0f} / image")
