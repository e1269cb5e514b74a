import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model, synth_batch
dev = torch.device("cuda:0")
for bs in (1, 32):
    m = build_model("s", dev); x = synth_batch(bs, 640, 0, dev)
    with torch.no_grad():
        for _ in range(5): m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): m(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"bs={bs}: enqueue {(t1-t0)/20*1e3:.3f} ms/step, total {(t2-t0)/20*1e3:.3f} ms/step")
# CUDA graph capture of the whole forward
m = build_model("s", dev); x = synth_batch(32, 640, 0, dev)
with torch.no_grad():
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): m(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            y = m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        print(f"graph replay bs=32: {(time.perf_counter()-t0)/20*1e3:.3f} ms/step")
    except Exception as e:
        print("graph capture failed:", repr(e)[:300])
