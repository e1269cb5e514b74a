import sys, torch
sys.argv = ["x", "32", "none"]
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_bench.py")).read())
from lead_yolo_amd import capi
for cfg in (0, 414, 221, 121):
    capi.lib().ly_debug_set_gemm_cfg(cfg)
    print("cfg", cfg)
    gemm_case("det.p3", 80, 128, 18)
    gemm_case("det.p5", 20, 512, 18)
capi.lib().ly_debug_set_gemm_cfg(0)
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
