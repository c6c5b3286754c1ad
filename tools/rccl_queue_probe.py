#!/usr/bin/env python3
"""How an initialised RCCL communicator in the same process changes the step time of the engine (one GPU, world size 1).
ORDER=pg_first|engine_first|none, EAGER=1 (communicator created at init_process_group) or 0 (by a first all_reduce),
MDX_STREAMS=1 for the one-stream mode, GPU_MAX_HW_QUEUES from the environment."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
if os.environ.get("IMPORT_LATE") != "1":      # IMPORT_LATE=1: libmdx.so is loaded after the communicator exists (as bench.py did)
    import mdx  # noqa: F401
    from mdx import synth, engine as eng

order = os.environ.get("ORDER", "pg_first")
eager = os.environ.get("EAGER", "1") == "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)


def start_pg():
    if eager:
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("nccl")
    if os.environ.get("NOCOLL") != "1":       # NOCOLL=1: no collective after the initialisation
        t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()


def make_engine():
    global synth, eng
    if os.environ.get("IMPORT_LATE") == "1":
        import mdx  # noqa: F401
        from mdx import synth, engine as eng
    s = synth.pseudo_complex(d=12.5)
    return eng.Engine(s, rc=1.0, dt=0.002, nstlist=20, nstcalcenergy=10, tc_groups=["Protein", "Non-Protein"], ref_t=[300, 300],
                      tau_t=[0.1, 0.1], nsttcouple=10, pcoupl=2, tau_p=2.0, ref_p=1.0, compressibility=4.5e-5, nstpcouple=10, dispcorr=2)


if order == "pg_first":
    start_pg(); e = make_engine()
elif order == "engine_first":
    e = make_engine(); e.step(20); e.synchronize(); start_pg()
else:
    e = make_engine()
if os.environ.get("ENGINES") == "3":      # the bench creates and closes two engines ahead of the timed one
    for _ in range(2):
        e.step(50); e.synchronize(); e.close(); e = make_engine()
e.step(1500); e.synchronize()
if os.environ.get("TIMING") == "1":
    e.enable_kernel_timing(True, stride=7)
if os.environ.get("BARRIER") == "1" and dist.is_initialized():
    dist.barrier()
if os.environ.get("TSYNC") == "1":
    torch.cuda.synchronize()
t0 = time.perf_counter(); e.step(2000); e.synchronize()
print("NOCOLL=%s IMPORT_LATE=%s " % (os.environ.get("NOCOLL", "0"), os.environ.get("IMPORT_LATE", "0")) + "ORDER=%s EAGER=%d ENGINES=%s TIMING=%s BARRIER=%s MDX_STREAMS=%s GPU_MAX_HW_QUEUES=%s: %.1f us/step" % (
    order, eager, os.environ.get("ENGINES", "1"), os.environ.get("TIMING", "0"), os.environ.get("BARRIER", "0"), os.environ.get("MDX_STREAMS", "3"), os.environ.get("GPU_MAX_HW_QUEUES", "default"), (time.perf_counter() - t0) / 2000 * 1e6), flush=True)
e.close()
if dist.is_initialized():
    dist.destroy_process_group()
