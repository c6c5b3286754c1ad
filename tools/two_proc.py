#!/usr/bin/env python3
"""Two engine processes stepping at the same time on one GPU (started together through a gloo barrier)."""
import os, sys, faulthandler, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(int(os.environ.get("LIMIT", "40")), exit=True)
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29545")
rank = int(os.environ.get("FAKE_RANK", "0"))
dist.init_process_group("gloo", rank=rank, world_size=2)
import mdx
from mdx import synth, engine as eng
s = synth.water_box(d=7.0)
e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=10, nstcalcenergy=0, nstcomm=0)
for kv in filter(None, os.environ.get("MDX_DEBUG", "").split(",")):
    e.debug_set(int(kv.split("=")[0]), int(kv.split("=")[1], 0))
dist.barrier()
n = int(os.environ.get("NSTEPS", "300"))
t0 = time.time()
for k in range(n // 10):
    e.step(10); e.synchronize(); print("rank %d: %d steps %.2f s" % (rank, 10 * (k + 1), time.time() - t0), flush=True)
print("rank %d: %d steps in %.2f s, streams now %d" % (rank, n, time.time() - t0, e.list_stats()["streams"]), flush=True)
