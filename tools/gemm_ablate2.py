import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["x", "32", "none"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_bench.py")).read())
from lead_yolo_amd import capi
for dbg in (15, 0):
    capi.lib().ly_debug_set_gemm(dbg)
    print("dbg", dbg)
    for b in (1, 4, 8, 16, 32, 64):
        B = b
        gemm_case(f"B={b} L16.cv3", 80, 128, 128)
capi.lib().ly_debug_set_gemm(0)
