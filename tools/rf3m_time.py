"""Runs the RFCBAMConv k=3 eval forward (bf16) a few times at the lead-yolo-s layer shapes (bs=64) — the target of rocprofv3 passes.
   python tools/rf3m_time.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L                                   # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for ci, co, s, shape in [(128, 128, 2, (64, 128, 80, 80)), (256, 256, 2, (64, 256, 40, 40))]:
    m = L.RFCBAMConv(ci, co, 3, s).to(dev).eval().bfloat16()
    xd = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(reps):
            m(xd)
    torch.cuda.synchronize()
print("done")
