import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lead_yolo_amd as L
from lead_yolo_amd import ops
dev = torch.device("cuda:0")
c, hw = 128, 80
m = L.RFCBAMConv(c, c, 3, 2).to(dev).eval()
x = torch.randn(32, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
P = m._packed(); th, tw = ops.pick_tile(hw // 2, hw // 2)
for _ in range(5):
    ops.rfcbam_stats(x, c, 32, hw, hw, c, 3, 2, wg=P['wq_stats'], th=th, tw=tw)
torch.cuda.synchronize()
