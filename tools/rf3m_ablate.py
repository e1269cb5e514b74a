"""Timing-only ablations of ly_rf3m_stats (LY_RM_SDBG = 1 no pooling sums | 2 no channel reductions | 4 no LDS reads in the loop | 8 no chunk boundary;
results are wrong) and of ly_rf3m_fwd (LY_RM_DBG = 1 no weight copies | 2 x prefetch of chunk 0 only | 4 no barrier): python tools/rf3m_ablate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L                                   # noqa: E402
from lead_yolo_amd import ops                               # noqa: E402
from oracle import synth                                    # noqa: E402

dev = torch.device("cuda:0")
ops.RF3M_MIN_UNITS = 0
for ci, co, s, shape in [(128, 128, 2, (64, 128, 80, 80)), (256, 256, 2, (64, 256, 40, 40))]:
    m = L.RFCBAMConv(ci, co, 3, s)
    m.load_state_dict(synth.synth_state(synth.shapes_of(m.state_dict()), 3), strict=True)
    m = m.to(dev).eval().bfloat16()
    xd = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(2):
            m(xd)
        torch.cuda.synchronize()
        ops.PROFILE = []
        for _ in range(9):
            m(xd)
        torch.cuda.synchronize()
        per = {}
        for r in ops.PROFILE:
            per.setdefault(r[0].split("<")[0], []).append(r[3].elapsed_time(r[4]) * 1e3)
        ops.PROFILE = None
    print(f"SDBG {os.environ.get('LY_RM_SDBG', '0')} DBG {os.environ.get('LY_RM_DBG', '0')} C={ci} O={co}: " +
          " | ".join(f"{k} {sorted(v)[len(v) // 2]:.1f} us" for k, v in per.items()), flush=True)
