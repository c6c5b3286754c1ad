"""Replica ensemble plumbing on CPU ranks: two processes over gloo stand in for two GPUs over RCCL.
Each rank produces the work curve of its own replica; one all-gather hands every rank all curves;
the Jarzynski estimate is identical on every rank and equals the single-process evaluation."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import mdx  # noqa: F401
    from mdx import ensemble
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seed = ensemble.replica_seed(20250829, rank)
        rng = np.random.default_rng(seed)
        # constant-force pull: W(t) = -k (r(t) - r(0)), a noisy approach curve per replica
        r = 3.0 - 0.3 * np.linspace(0, 1, 50) + 0.02 * rng.normal(size=50).cumsum()
        local = -500.0 * (r - r[0])
        allw = ensemble.gather_work(local)
        dF = ensemble.jarzynski_free_energy(allw, 300.0)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), local=local, allw=allw, dF=dF, seed=np.array([seed]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_and_jarzynski(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert r0["seed"][0] != r1["seed"][0]
    assert r0["allw"].shape == (2, 50)
    # every rank holds the same gathered matrix, rows ordered by rank
    assert np.array_equal(r0["allw"], r1["allw"])
    assert np.array_equal(r0["allw"][0], r0["local"]) and np.array_equal(r0["allw"][1], r1["local"])
    assert np.array_equal(r0["dF"], r1["dF"])
    sys.path.insert(0, ROOT)
    import mdx  # noqa: F401
    from mdx import ensemble
    assert np.allclose(r0["dF"], ensemble.jarzynski_free_energy(np.stack([r0["local"], r1["local"]]), 300.0))


def test_jarzynski_limits():
    sys.path.insert(0, ROOT)
    import mdx  # noqa: F401
    from mdx import ensemble, units as U
    w = np.tile(np.linspace(0, -50, 11), (8, 1))
    # identical replicas: dF = W
    assert np.allclose(ensemble.jarzynski_free_energy(w), w[0])
    # Gaussian work distribution: dF = <W> - var/(2kT)
    rng = np.random.default_rng(0)
    kT = U.BOLTZ * 300.0
    ws = rng.normal(loc=-20.0, scale=1.0, size=(200000, 1))
    assert abs(ensemble.jarzynski_free_energy(ws)[0] - (-20.0 - 1.0 / (2 * kT))) < 0.02
    # without a process group the gather is the identity
    assert ensemble.gather_work(np.arange(4.0)).shape == (1, 4)
    # numerically safe for work values far beyond exp() range
    big = np.array([[-5000.0], [-5001.0]])
    assert np.isfinite(ensemble.jarzynski_free_energy(big)).all()
