import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["x", "32", "none"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_bench.py")).read())
from lead_yolo_amd import capi
for cfg in (0, 14, 24, 44, 22, 42, 41):
    capi.lib().ly_debug_set_conv3_cfg(cfg)
    print("cfg", cfg)
    conv_case("L16", 80, 64, 64)
    conv_case("L12/19", 40, 128, 128)
    conv_case("L22", 20, 256, 256)
capi.lib().ly_debug_set_conv3_cfg(0)
