import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["x", "32", "none"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_bench.py")).read())
for b in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    B = b
    gemm_case(f"B={b} L12.cv3", 40, 256, 256)
