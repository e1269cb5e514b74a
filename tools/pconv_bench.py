import sys, os
sys.path.insert(0, "/root/repo")
import torch
import lead_yolo_amd as L
from lead_yolo_amd import ops, pack
dev = torch.device("cuda:0")
B = 64
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for c, hw in ((24, 160), (40, 80), (80, 40), (160, 20)):
    c4 = c // 4; c4p = (c4 + 3) // 4 * 4; c4q = (c4 + 31) // 32 * 32
    M = B * hw * hw
    x = torch.randn(B, c, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(c4, c4, 3, 3, device=dev)
    wp = pack.frag_pack3(pack.conv_taps_matrix(w, c4q), planes=1)
    z = x.clone()
    xr, ld = ops.rows(x)
    zr, _ = ops.rows(z)
    us = t(lambda: ops.conv3x3(M=M, H=hw, W=hw, Cin=c4p, N=c4, x=xr, ldx=c, wp=wp, out=zr, ldo=c))
    print(f"pconv C={c} c4={c4} {hw}x{hw}: {us:.1f} us  ({M * 2 * (c4p + c4) * 2 / us / 1e3:.0f} GB/s useful)")
