import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["x", "32", "none"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_bench.py")).read())
from lead_yolo_amd import capi
B = 32
for dbg, what in ((0, "full"), (16, "no weight loads"), (2, "no LDS reads + MFMA"), (2 + 16, "no MFMA, no weights"), (4, "no prefetch"), (1, "no commit"), (8, "no stores"),
                  (4 + 1, "no prefetch/commit"), (4 + 1 + 16, "no prefetch/commit/weights"), (4 + 1 + 16 + 8, "MFMA + LDS reads only")):
    capi.lib().ly_debug_set_gemm(dbg)
    print("dbg", dbg, what)
    gemm_case("L12.cv3", 40, 256, 256)
    gemm_case("L16.cv3", 80, 128, 128)
capi.lib().ly_debug_set_gemm(0)
