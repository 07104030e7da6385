"""One-rank RCCL sanity check: process-group init on the nccl backend and the ReduceOp.AVG all-reduce the data-parallel path uses."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
g = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
ref = g.clone()
dist.all_reduce(g, op=dist.ReduceOp.AVG)
dist.broadcast(g, 0)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(g, ref)
print("RCCL", torch.cuda.nccl.version(), "ReduceOp.AVG ok")
dist.destroy_process_group()
