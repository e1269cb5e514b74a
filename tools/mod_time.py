"""Device time per call of the north-star modules, REP calls of one module captured back to back in ONE hipGraph (no replay floor,
no host launch cost in the figure).  Dev tool, GPU box.
    python tools/mod_time.py [batch=64] [mlp|rf|all] [bf16|f32] [train|eval] [norf3m]
`train`: the training-mode forward pieces of a BasicStage (statistics pass + forward) through the module in train() under no_grad."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
which = sys.argv[2] if len(sys.argv) > 2 else "all"
DT = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.bfloat16
TRAIN = len(sys.argv) > 4 and sys.argv[4] == "train"
if "norf3m" in sys.argv:                  # RFCBAMConv k=3 eval on the lane = channel kernels at every size
    from lead_yolo_amd import modules as _m
    _m.RF3M = False
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


def case(name, mod, shape):
    m = mod.to(dev)
    m = m.train() if TRAIN else m.eval()
    x = torch.randn(*shape, device=dev).to(DT).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = m(x)
        us = graph_time(lambda: m(x))
    nbytes = (x.numel() + y.numel()) * x.element_size()
    print(f"{name:<40} {us:8.1f} us   {nbytes / us / 1e3:8.1f} GB/s", flush=True)


if which in ("all", "mlp"):
    for c, hw in ((24, 160), (40, 80), (80, 40), (160, 20)):
        case(f"mlpblock C={c} {hw}x{hw}", L.BasicStage(c, 1), (B, c, hw, hw))
if which in ("all", "rf"):
    case("rfcbam L9  160->256 k1 20x20", L.RFCBAMConv(160, 256, 1, 1), (B, 160, 20, 20))
    case("rfcbam L13 256->128 k1 40x40", L.RFCBAMConv(256, 128, 1, 1), (B, 256, 40, 40))
    case("rfcbam L17 128->128 k3s2 80x80", L.RFCBAMConv(128, 128, 3, 2), (B, 128, 80, 80))
    case("rfcbam L20 256->256 k3s2 40x40", L.RFCBAMConv(256, 256, 3, 2), (B, 256, 40, 40))
