import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
torch.manual_seed(0)
dev = "cuda"
def probe(name, m, shape, eps=1e-4):
    m = m.to(dev).train()
    for mm in m.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.eps, mm.momentum = 1e-3, 0.03
    x = torch.randn(*shape, device=dev).contiguous(memory_format=torch.channels_last)
    res = []
    r = None
    for it in range(2):
        m.zero_grad()
        xt = x.clone().requires_grad_(True)
        y = m(xt)
        if r is None:
            r = torch.randn_like(y)
            rr = r
        else:
            rr = r * (1 + eps * torch.randn_like(r))
        (y * rr).sum().backward()
        res.append({"dx": xt.grad.clone(), **{k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}})
    amp = {k: float((res[1][k] - res[0][k]).abs().max() / (res[0][k].abs().max() + 1e-30)) / eps for k in res[0]}
    print(name, shape, "amplification:", {k: round(v, 1) for k, v in amp.items() if v > 20} or "ok (<20x)", " dx", round(amp["dx"], 1))
probe("rfcbam k3s2", L.RFCBAMConv(256, 256, 3, 2), (4, 256, 20, 20))
probe("rfcbam k3s2", L.RFCBAMConv(128, 128, 3, 2), (4, 128, 40, 40))
probe("rfcbam k1", L.RFCBAMConv(160, 256, 1, 1), (4, 160, 10, 10))
probe("c3ca", L.C3_CA(256, 256, 1, False), (4, 256, 20, 20))
probe("sppf", L.SPPF(160, 160, 5), (4, 160, 10, 10))
probe("basicstage", L.BasicStage(80, 1), (4, 80, 20, 20))
