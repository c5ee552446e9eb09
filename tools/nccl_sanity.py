"""RCCL sanity check used before trusting bench.py's multi-rank path: one rank, nccl backend, the same calls
(init with device_id, barrier, MAX all-reduce of a device tensor, object all-gather of the slab layout, the in-place
all_gather_into_tensor on the padded vector, the asynchronous in-place broadcasts of the overlapped form)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import dto_amd

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
D = dto_amd.distributed
layout = D.slab_layout(0, 1000)
buf, full = D.alloc_gather_vector(1000, layout, torch.float64, torch.device("cuda", 0))
full.copy_(torch.arange(1000, dtype=torch.float64, device="cuda:0"))
D.gather_slabs_inplace(full, layout, buffer=buf)
for w in D.gather_slabs_async(full, layout):
    w.wait()
torch.cuda.synchronize()
assert buf is not None and float(full.sum().item()) == 999 * 1000 / 2
print("nccl ok", float(t.item()), layout)
dist.destroy_process_group()
