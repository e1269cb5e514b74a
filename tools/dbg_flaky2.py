import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
torch.manual_seed(0)
dev = "cuda"
def check(name, m, shape, iters=20, multi=False):
    m = m.to(dev).train()
    for mm in m.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.eps, mm.momentum = 1e-3, 0.03
    x = torch.randn(*shape, device=dev).contiguous(memory_format=torch.channels_last)
    ref, worst = None, {}
    for it in range(iters):
        m.zero_grad()
        xt = x.clone().requires_grad_(True)
        y = m([xt] if multi else xt)
        y = y[0] if multi else y
        if it == 0:
            r = torch.randn_like(y)
        (y * r).sum().backward()
        cur = {"dx": xt.grad.clone(), **{k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}}
        if ref is None:
            ref = cur
        else:
            for k in cur:
                d = float((cur[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30))
                worst[k] = max(worst.get(k, 0), d)
    bad = {k: f"{v:.1e}" for k, v in worst.items() if v > 2e-5}
    print(name, shape, "max dev", f"{max(worst.values()):.1e}", bad)
check("rfcbam k3s2", L.RFCBAMConv(256, 256, 3, 2), (4, 256, 8, 8))
check("rfcbam k3s2", L.RFCBAMConv(128, 128, 3, 2), (4, 128, 16, 16))
check("rfcbam k1", L.RFCBAMConv(160, 256, 1, 1), (4, 160, 4, 4))
check("rfcbam k1", L.RFCBAMConv(256, 128, 1, 1), (4, 256, 8, 8))
check("c3ca", L.C3_CA(256, 256, 1, False), (4, 256, 8, 8))
check("c3ca", L.C3_CA(512, 512, 1, False), (4, 512, 4, 4))
check("sppf", L.SPPF(160, 160, 5), (4, 160, 4, 4))
det = L.Detect(1, ((10, 13, 16, 30, 33, 23),), (256,))
det.stride = torch.tensor([16.0])
check("detect", det, (4, 256, 8, 8), multi=True)
check("patchmerge", L.PatchMerging_FasterNet(40, 80, 2, 2), (4, 40, 16, 16))
