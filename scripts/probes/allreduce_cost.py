"""host-side cost of one torch.distributed all_reduce call (single rank, nccl) as the solve loop issues it"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from dsopp_amd import distributed
stream = torch.cuda.Stream()
t = torch.zeros(6276, dtype=torch.float64, device="cuda")
fn = distributed.make_device_allreduce(dist, torch, stream, 0)
ptr = t.data_ptr()
for _ in range(50): fn(ptr, 6276, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): fn(ptr, 6276, 0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host time per call {1e6*(t1-t0)/2000:.1f} us; incl. drain {1e6*(t2-t0)/2000:.1f} us")
dist.destroy_process_group()
