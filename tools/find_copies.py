import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
model = B.build_model("s", dev)
x = B.synth_batch(32, 640, 0, dev)
with torch.no_grad():
    for _ in range(3): model(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model(x)
        torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60, max_src_column_width=100))
