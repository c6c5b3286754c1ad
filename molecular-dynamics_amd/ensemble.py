"""Replica ensembles: one SMD replica per GPU/process, pull work all-gathered for Jarzynski
averaging. The only collective on the path is one all_gather of the per-replica work curve
(a few thousand floats): the replicas themselves never exchange data.

Covers the pipeline's per-pose SMD fan-out (/root/reference/Snakefile:753-802 runs one
`short_smd_pose` per docking pose) turned into an ensemble of independent pulls.
"""
from __future__ import annotations

import numpy as np

from . import units as U


def jarzynski_free_energy(work: np.ndarray, temperature: float = 300.0) -> np.ndarray:
    """dF(t) = -kT ln < exp(-W(t)/kT) > over replicas; `work` is (n_replicas, n_samples)."""
    kT = U.BOLTZ * temperature
    w = np.asarray(work, dtype=np.float64)
    a = -w / kT
    m = a.max(axis=0)
    return -kT * (m + np.log(np.exp(a - m).mean(axis=0)))


def gather_work(local_work: np.ndarray, group=None) -> np.ndarray:
    """All-gather equally sized work curves from every rank (RCCL on GPU ranks, gloo on CPU).
    Returns (world_size, n_samples). Without an initialised process group it is the identity."""
    import torch
    import torch.distributed as dist
    w = np.ascontiguousarray(local_work, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return w[None, :]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.from_numpy(w).to(dev)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return torch.stack(out).cpu().numpy()


def replica_seed(base_seed: int, rank: int) -> int:
    """Distinct, reproducible stochastic-coupling seed per replica."""
    return (int(base_seed) * 1000003 + 7919 * (int(rank) + 1)) & 0x7FFFFFFFFFFFFFFF
