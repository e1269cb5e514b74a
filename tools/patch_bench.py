"""PatchEmbed (4x4 stride-4 patches of the NCHW image -> GEMM) alone, at the north-star image size: input dtype x output width.  Dev tool.
    python tools/patch_bench.py [bs=32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L

dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for cout in (32, 40, 64, 80):
    m = L.PatchEmbed_FasterNet(3, cout, 4, 4).to(dev).eval()
    for idt in (torch.float32, torch.bfloat16, torch.uint8):
        xs = [(torch.rand(bs, 3, 640, 640, device=dev) * 255).to(idt) if idt == torch.uint8 else torch.rand(bs, 3, 640, 640, device=dev).to(idt) for _ in range(4)]
        i = [0]

        def run():
            i[0] += 1
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                return m(xs[i[0] % 4])
        us = timeit(run)
        by = xs[0].numel() * xs[0].element_size() + bs * 160 * 160 * cout * 2
        print(f"cout={cout:3d} image {str(idt):<15} {us:7.1f} us  {by / us / 1e6:6.2f} TB/s")
