import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["x", "32", "none"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_bench.py")).read())
from lead_yolo_amd import capi, ops
ops.PROFILE = None
for v, what in ((8, "MT=4 for N>128 (one cout group)"), (0, "MT=2, two cout groups")):
    capi.lib().ly_debug_set_rf3(v)
    print(what)
    module_case("rfcbam L20 256->256 k3s2 40x40", L.RFCBAMConv(256, 256, 3, 2), (B, 256, 40, 40), 2.0 * B * 400 * 9 * 256 * 256)
capi.lib().ly_debug_set_rf3(0)
