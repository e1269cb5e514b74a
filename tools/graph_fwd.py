import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
dev = torch.device("cuda:0")
model = B.build_model("s", dev)
x = B.synth_batch(32, 640, 0, dev)
with torch.no_grad():
    for _ in range(5): model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): model(x)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 30
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): model(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = model(x)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 30
    ref = model(x)
    g.replay(); torch.cuda.synchronize()
    print("eager %.3f ms  graph %.3f ms  max diff %.2e" % (eager * 1e3, graph * 1e3, float((out[0] - ref[0]).abs().max())))
