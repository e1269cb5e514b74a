"""Graph replay time of the eval forward vs the number of concurrent sub-batch streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import lead_yolo_amd as L
dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = B.build_model("s", dev)
x = B.synth_batch(bs, 640, 0, dev)
for parts in (1, 2, 4, 8, 16):
    if bs % parts:
        continue
    g = L.GraphedForward(model, x, parts=parts)
    for _ in range(5): g()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): g()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print(f"bs={bs} parts={parts}: {dt*1e3:.3f} ms/step  {bs/dt:.0f} img/s")
    del g
