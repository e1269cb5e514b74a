"""hipGraph capture while an RCCL process group (and its watchdog thread) is alive, as in the multi-GPU bench."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); dist.barrier(device_ids=[0])
import bench as B, lead_yolo_amd as L
model = B.build_model("s", torch.device("cuda:0"))
x = B.synth_batch(32, 640, 0, torch.device("cuda:0"))
g = L.GraphedForward(model, x)
for _ in range(5): g()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g()
dist.barrier(device_ids=[0]); torch.cuda.synchronize()
print("graph replay with live process group: %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
dist.destroy_process_group()
