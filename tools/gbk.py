import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["x", "32", "none"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_bench.py")).read())
from lead_yolo_amd import capi
for bk in (64, 128):
    capi.lib().ly_debug_set_gemm_bk(bk)
    print("bk", bk)
    gemm_case("L12.cv12", 40, 336, 256)
    gemm_case("L12.cv3", 40, 256, 256)
    gemm_case("L16.cv3", 80, 128, 128)
    gemm_case("L16.cv12", 80, 168, 128)
    gemm_case("L22.cv12", 20, 512, 512)
    gemm_case("L12.m.cv1", 40, 128, 128)
    gemm_case("L19.cv12", 40, 256, 256)
    gemm_case("L9 160->256", 20, 160, 256)
capi.lib().ly_debug_set_gemm_bk(0)
