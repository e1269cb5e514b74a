import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["x", "32", "none"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_bench.py")).read())
from lead_yolo_amd import capi
for dbg in (0, 2, 8, 10):
    capi.lib().ly_debug_set_conv3(dbg)
    print(f"dbg={dbg:2d} skip mfma={(dbg>>1)&1} commit={(dbg>>3)&1}")
    conv_case("L12/19", 40, 128, 128)
    conv_case("L16", 80, 64, 64)
capi.lib().ly_debug_set_conv3(0)
