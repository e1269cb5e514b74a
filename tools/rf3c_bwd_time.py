"""one training forward + backward of RFCBAMConv k=3 at the lead-yolo-s layer shapes (bs=64, bf16): the target of rocprofv3 passes"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L                                   # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for ci, co, shape in [(128, 128, (64, 128, 80, 80)), (256, 256, (64, 256, 40, 40))]:
    m = L.RFCBAMConv(ci, co, 3, 2).to(dev).train()
    xd = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    dy = None
    for _ in range(reps):
        y = m(xd)
        if dy is None:
            dy = torch.randn_like(y)
        y.backward(dy)
    torch.cuda.synchronize()
print("done")
