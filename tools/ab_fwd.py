"""A/B of a debug switch on the whole forward: eager and graph-replay step time.  usage: ab_fwd.py <setter> <v0> <v1> ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import lead_yolo_amd as L
from lead_yolo_amd import capi
setter = sys.argv[1]
vals = [int(v) for v in sys.argv[2:]]
dev = torch.device("cuda:0")
model = B.build_model("s", dev)
x = B.synth_batch(32, 640, 0, dev)
def t(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    for v in vals:
        getattr(capi.lib(), setter)(v)
        with torch.no_grad():
            e = t(lambda: model(x))
        g = L.GraphedForward(model, x)
        gr = t(g)
        del g
        print(f"{setter}({v}): eager {e:.3f} ms  graph {gr:.3f} ms")
