"""Per-kernel micro-benchmarks on the real lead-yolo-s layer shapes.  Dev tool.
    python tools/kernel_bench.py [batch=32] [all|gemm|conv|mlp|rf|c3|patch|sweep] [f32|bf16]
(`sweep`: the L12.cv3 GEMM over batch 1..256 — fixed vs per-tile cost)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
from lead_yolo_amd import ops, pack

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
DT = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == "bf16") else torch.float32
ES = 2 if DT == torch.bfloat16 else 4
PL = 1 if DT == torch.bfloat16 else 2


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
    return e0.elapsed_time(e1) / iters * 1e3  # us


def report(name, us, flops, bytes_):
    print(f"{name:<44} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s  {bytes_ / us / 1e3:8.1f} GB/s")


def gemm_case(name, hw, k, n):
    M = B * hw * hw
    a = torch.randn(M, k, device=dev).to(DT)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    wp = pack.frag_pack3(w, planes=PL)
    out = torch.empty(M, n, device=dev, dtype=DT)
    sc = torch.ones(n, device=dev)
    sh = torch.zeros(n, device=dev)
    fn = lambda: ops.gemm(M=M, H=hw, W=hw, K=k, N=n, a0=a, lda0=k, k0=k, wp=wp, out=out, ldo=n, e_scale=sc, e_shift=sh, act=2)
    report(f"gemm {name} M={M} K={k} N={n}", timeit(fn), 2.0 * M * k * n, 1.0 * ES * M * (k + n))


def conv_case(name, hw, c, n):
    M = B * hw * hw
    x = torch.randn(M, c, device=dev).to(DT)
    w = torch.randn(n, c, 3, 3, device=dev) / (9 * c) ** 0.5
    wp = pack.frag_pack3(pack.conv_taps_matrix(w, 32), planes=PL)
    out = torch.empty(M, n, device=dev, dtype=DT)
    sc = torch.ones(n, device=dev)
    sh = torch.zeros(n, device=dev)
    fn = lambda: ops.conv3x3(M=M, H=hw, W=hw, Cin=c, N=n, x=x, ldx=c, wp=wp, out=out, ldo=n, e_scale=sc, e_shift=sh, act=2)
    report(f"conv3x3 {name} M={M} C={c} N={n}", timeit(fn), 2.0 * M * 9 * c * n, 1.0 * ES * M * (c + n))


def module_case(name, mod, shape, flops):
    m = mod.to(dev).eval()
    x = torch.randn(*shape, device=dev).to(DT).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = m(x)
        us = timeit(lambda: m(x))
    report(name, us, flops, 1.0 * ES * (x.numel() + y.numel()))


which = sys.argv[2] if len(sys.argv) > 2 else "all"
if which in ("all", "gemm"):
    gemm_case("L12.cv12", 40, 336, 256)
    gemm_case("L12.cv3", 40, 256, 256)
    gemm_case("L12.m.cv1", 40, 128, 128)
    gemm_case("L16.cv12", 80, 168, 128)
    gemm_case("L16.m.cv1", 80, 64, 64)
    gemm_case("L16.cv3", 80, 128, 128)
    gemm_case("L22.cv12", 20, 512, 512)
    gemm_case("L22.m.cv1", 20, 256, 256)
    gemm_case("L13.conv", 40, 256, 128)
    gemm_case("L9.conv", 20, 160, 256)
    gemm_case("L8.sppf.cv2", 20, 320, 160)
    gemm_case("det.p3", 80, 128, 18)
if which in ("all", "conv"):
    conv_case("L16", 80, 64, 64)
    conv_case("L12/19", 40, 128, 128)
    conv_case("L22", 20, 256, 256)
if which in ("all", "mlp"):
    for c, hw in ((24, 160), (40, 80), (80, 40), (160, 20)):
        module_case(f"mlpblock C={c} {hw}x{hw}", L.BasicStage(c, 1), (B, c, hw, hw), 2.0 * B * hw * hw * (9 * (c // 4) ** 2 + 4 * c * c))
if which in ("all", "rf"):
    module_case("rfcbam L9  160->256 k1 20x20", L.RFCBAMConv(160, 256, 1, 1), (B, 160, 20, 20), 2.0 * B * 400 * 160 * 256)
    module_case("rfcbam L13 256->128 k1 40x40", L.RFCBAMConv(256, 128, 1, 1), (B, 256, 40, 40), 2.0 * B * 1600 * 256 * 128)
    module_case("rfcbam L17 128->128 k3s2 80x80", L.RFCBAMConv(128, 128, 3, 2), (B, 128, 80, 80), 2.0 * B * 1600 * 9 * 128 * 128)
    module_case("rfcbam L20 256->256 k3s2 40x40", L.RFCBAMConv(256, 256, 3, 2), (B, 256, 40, 40), 2.0 * B * 400 * 9 * 256 * 256)
if which in ("all", "c3"):
    module_case("c3ca L16 168->128 80x80", L.C3_CA(168, 128, 1, False), (B, 168, 80, 80), 2.0 * B * 6400 * (168 * 128 + 64 * 64 * 10 + 128 * 128))
    module_case("c3ca L12 336->256 40x40", L.C3_CA(336, 256, 1, False), (B, 336, 40, 40), 2.0 * B * 1600 * (336 * 256 + 128 * 128 * 10 + 256 * 256))
    module_case("c3ca L22 512->512 20x20", L.C3_CA(512, 512, 1, False), (B, 512, 20, 20), 2.0 * B * 400 * (512 * 512 + 256 * 256 * 10 + 512 * 512))
if which in ("all", "patch"):
    module_case("patchembed 3->24 640", L.PatchEmbed_FasterNet(3, 24, 4, 4), (B, 3, 640, 640), 2.0 * B * 25600 * 48 * 24)
    module_case("patchmerge 24->40 160", L.PatchMerging_FasterNet(24, 40, 2, 2), (B, 24, 160, 160), 2.0 * B * 6400 * 96 * 40)

if which == "sweep":
    for b in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        B = b
        gemm_case(f"B={b} L12.cv3", 40, 256, 256)
