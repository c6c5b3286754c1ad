#!/usr/bin/env python3
"""Diagnostic: bench.py's main() under an eagerly created RCCL communicator with single ingredients taken out
(SKIP=barrier | timing | sync), to find which one interacts with the communicator."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
skip = os.environ.get("SKIP", "")
if "barrier" in skip:
    dist.barrier = lambda *a, **k: None
if "sync" in skip:
    torch.cuda.synchronize = lambda *a, **k: None
import bench
if "timing" in skip:
    import mdx  # noqa: F401
    from mdx import engine as eng
    eng.Engine.enable_kernel_timing = lambda self, on=True, stride=7: None
    eng.Engine.kernel_timing = lambda self: (1.0, 1)
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-output", "--steps", "2000", "--warmup", "100"]
bench.main()
