"""RCCL sanity check used before trusting bench.py's multi-rank path: one rank, nccl backend, the same calls
(init with device_id, barrier, MAX all-reduce of a device tensor)."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print("nccl ok", float(t.item()))
dist.destroy_process_group()
