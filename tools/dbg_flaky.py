import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
from lead_yolo_amd import ops, pack
torch.manual_seed(0)
dev = "cuda"
for C, hw in ((40, 32), (80, 16), (24, 64), (160, 8)):
    m = L.BasicStage(C, 1).to(dev).train()
    x = torch.randn(4, C, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    r = torch.randn(4, C, hw, hw, device=dev)
    ref = None
    worst = {}
    for it in range(30):
        m.zero_grad()
        xt = x.clone().requires_grad_(True)
        y = m(xt)
        (y * r).sum().backward()
        cur = {"dx": xt.grad.clone(), **{k: p.grad.clone() for k, p in m.named_parameters()}}
        if ref is None:
            ref = cur
        else:
            for k in cur:
                d = float((cur[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30))
                worst[k] = max(worst.get(k, 0), d)
    print(C, {k.split("blocks.0.")[-1]: f"{v:.1e}" for k, v in worst.items()})
# isolate the contraction kernels at the MLP-backward shapes
for C, hw in ((40, 32), (80, 16)):
    n = 4; m_ = n * hw * hw
    z = torch.randn(n, C, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w1 = torch.randn(2 * C, C, device=dev)
    pk = pack.frag_pack3(w1)
    outs = []
    for it in range(30):
        u = ops.empty_nhwc(n, 2 * C, hw, hw, z)
        u.fill_(float("nan"))
        ops.gemm(M=m_, H=hw, W=hw, K=C, N=2 * C, a0=z, lda0=C, k0=C, wp=pk, out=u, ldo=2 * C)
        outs.append(u)
    print("gemm K=%d N=%d max dev" % (C, 2 * C), max(float((o - outs[0]).abs().max()) for o in outs), "nan", any(bool(torch.isnan(o).any()) for o in outs))
    dy = torch.randn(n, 2 * C, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    pk2 = pack.frag_pack3(w1.t().contiguous())
    outs = []
    for it in range(30):
        g = ops.empty_nhwc(n, C, hw, hw, z); g.fill_(float("nan"))
        ops.gemm(M=m_, H=hw, W=hw, K=2 * C, N=C, a0=dy, lda0=2 * C, k0=2 * C, wp=pk2, out=g, ldo=C)
        outs.append(g)
    print("gemm K=%d N=%d max dev" % (2 * C, C), max(float((o - outs[0]).abs().max()) for o in outs), "nan", any(bool(torch.isnan(o).any()) for o in outs))
    c4 = C // 4; c4p = (c4 + 3) // 4 * 4
    wpc = torch.randn(c4, c4, 3, 3, device=dev)
    pk3 = pack.frag_pack3(pack.conv_taps_matrix(wpc, 32))
    outs = []
    for it in range(30):
        zz = z.clone()
        ops.conv3x3(M=m_, H=hw, W=hw, Cin=c4p, N=c4, x=z, ldx=C, wp=pk3, out=zz, ldo=C)
        outs.append(zz)
    print("conv3 Cin=%d N=%d max dev" % (c4p, c4), max(float((o - outs[0]).abs().max()) for o in outs))
