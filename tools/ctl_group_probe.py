#!/usr/bin/env python3
"""Two ranks (both on GPU 0, no RCCL collective issued): the `nccl` process group initialised without a device, a host-side
gloo group beside it, a barrier and a maximum over the ranks through that group: the bench's bracket around its timed region."""
import os
import time
import torch
import torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl")
ctl = dist.new_group(backend="gloo")
dist.barrier(group=ctl)
t = torch.tensor([1.0 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
assert float(t.item()) == float(world), t
dist.barrier(group=ctl)
if rank == 0:
    print("control group ok: world %d, max %.1f" % (world, float(t.item())), flush=True)
dist.destroy_process_group()
