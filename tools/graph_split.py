import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import lead_yolo_amd.modules as MM
MM._overlap = lambda: False        # the per-layer forks share one auxiliary stream: not combinable with several capture streams
dev = torch.device("cuda:0")
model = B.build_model("s", dev)
x = B.synth_batch(32, 640, 0, dev)
def run(parts):
    xs = list(x.chunk(parts, 0))
    streams = [torch.cuda.Stream() for _ in range(parts)]
    with torch.no_grad():
        for s, xi in zip(streams, xs):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2): model(xi)
        for s in streams: torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            outs = []
            for s, xi in zip(streams, xs):
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    outs.append(model(xi))
            for s in streams: main.wait_stream(s)
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
    print(f"parts={parts}: {dt*1e3:.3f} ms/step  {32/dt:.0f} img/s")
for parts in (1, 2, 4):
    run(parts)
