import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["x", "32", "none"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_bench.py")).read())
from lead_yolo_amd import capi
for dbg in (0, 1, 2, 4, 8, 3, 7, 15, 11):
    capi.lib().ly_debug_set_gemm(dbg)
    print(f"dbg={dbg:2d} skip commit={dbg&1} mfma={(dbg>>1)&1} prefetch={(dbg>>2)&1} store={(dbg>>3)&1}")
    gemm_case("L16.cv3", 80, 128, 128)
    gemm_case("L12.cv12", 40, 336, 256)
capi.lib().ly_debug_set_gemm(0)
