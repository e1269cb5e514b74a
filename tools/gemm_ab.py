import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["x", "32", "none"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_bench.py")).read())
from lead_yolo_amd import capi
B = 32
for d2 in (0, 1):
    capi.lib().ly_debug_set_gemm_d2(d2)
    print("two-deep prefetch", d2)
    gemm_case("L12.cv12", 40, 336, 256)
    gemm_case("L12.cv3", 40, 256, 256)
    gemm_case("L16.cv3", 80, 128, 128)
    gemm_case("L16.cv12", 80, 168, 128)
    gemm_case("L22.cv12", 20, 512, 512)
    gemm_case("L12.m.cv1", 40, 128, 128)
    gemm_case("L9 160->256", 20, 160, 256)
    gemm_case("L16.m.cv1", 80, 64, 64)
    gemm_case("L2 merge-like", 80, 96, 40)
capi.lib().ly_debug_set_gemm_d2(1)
