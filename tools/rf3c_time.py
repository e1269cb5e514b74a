"""Runs the lane = channel RFCBAMConv k=3 forward kernels alone at the lead-yolo-s layer shapes (bs=64) — the target of rocprofv3 passes.
   python tools/rf3c_time.py [bf16|f32] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L                                   # noqa: E402
from lead_yolo_amd import ops                               # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for ci, co, s, shape in [(128, 128, 2, (64, 128, 80, 80)), (256, 256, 2, (64, 256, 40, 40))]:
    m = L.RFCBAMConv(ci, co, 3, s).to(dev).eval()
    if dt == torch.bfloat16:
        m = m.bfloat16()
    xd = torch.randn(shape, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    xr, ld = ops.rows(xd)
    n, c, h, w = xr.shape
    ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
    P = m._packed(ops.planes_of(xr))
    th, tw = ops.pick_tile_c(ho, wo, s)
    wa, wb = m.se.fc[0].weight.detach().float().contiguous(), m.se.fc[2].weight.detach().float().contiguous()
    out = ops.empty_nhwc(n, co, ho, wo, xr)
    for _ in range(reps):
        mm, part = ops.rf3c_stats(xr, ld, n, h, w, c, s, P["wq_c"], th, tw)
        ca, rfa = ops.rfcbam_mid(part, h * w, wa, wb, m.se.ratio, mm, P["w18"])
        ops.rf3c_fwd(n=n, h=h, w=w, c=c, ho=ho, wo=wo, N=co, s=s, th=th, tw=tw, x=xr, ldx=ld, wq=P["wq_c"], ca=ca, rfa=rfa, wp=P["wp_c"], ldo=co,
                     out=out, e_scale=P["es"], e_shift=P["eb"])
    torch.cuda.synchronize()
print("done")
