import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lead_yolo_amd import ops, pack, capi
dev = torch.device("cuda:0")
hw, k, n, cfg = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dbg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
capi.lib().ly_debug_set_gemm_cfg(cfg)
capi.lib().ly_debug_set_gemm(dbg)
M = 32 * hw * hw
a = torch.randn(M, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5
wp = pack.frag_pack3(w); out = torch.empty(M, n, device=dev)
sc = torch.ones(n, device=dev); sh = torch.zeros(n, device=dev)
for _ in range(10):
    ops.gemm(M=M, H=hw, W=hw, K=k, N=n, a0=a, lda0=k, k0=k, wp=wp, out=out, ldo=n, e_scale=sc, e_shift=sh, act=2)
torch.cuda.synchronize()
