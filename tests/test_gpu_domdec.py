"""Spatial domain decomposition on the GPU: domains on the one device (one thread each, in-process transport; the
torch.distributed transport is tested on CPU) against the
single-handle run of the same water box. The reference has no decomposed runs (gmx mdrun -ntmpi 1,
/root/reference/Snakefile:232-233): the single handle, itself checked against the oracle, is the yardstick."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, box):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "domdec_ranks.py"), str(world), "12", str(box)],
                       capture_output=True, text=True, timeout=170)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("world,box", [(2, 9.0), (4, 9.0)])
def test_domains_match_single_handle(world, box):
    res = _run(world, box)
    part = res["traj"]["partition"]
    assert part["halo"] > 0 and part["home"] > 0
    # the mesh transform is shared by planes of the half-complex grid (reduce-scatter, own planes, all-gather; the
    # in-process transport fills the planes a domain does not own with NaN in between), energies included; the
    # "traj_gather" run below keeps the summed real-space grids
    assert part["mesh"] == "slabs" and res["coupled"]["mesh"] == "slabs" and res["traj_gather"]["partition"]["mesh"] == "replicated", res
    # forces through the whole decomposed path (halo exchange, summed charge grid): one step from rest. The same
    # measure between the one-domain decomposed path and the plain handle, which differ in atom order and padding
    # only, is 8e-6 (float32 evaluation order through the constrained update); two domains measure 1.0e-5.
    assert res["rest"]["dv_rel_rms"] < 2e-5, res
    # 12 steps with one renewal of the partition: 2.3e-7 nm RMS, the same as the one-domain path against the plain handle
    assert res["traj"]["dx_rms"] < 2e-6 and res["traj"]["dx_max"] < 5e-5, res
    # (the partition is renewed by migration: molecules that left a brick and the halo go to the ranks that need them;
    # the same run with every molecule gathered on every rank instead)
    assert res["traj_gather"]["dx_rms"] < 2e-6 and res["traj_gather"]["dx_max"] < 5e-5, res
    # Thermostat, centre-of-mass removal and energies across the domains (21 steps, coupling and energies every 10):
    # the partial sums of the domains add up to the single handle's energies, virial and pressure, and the one
    # V-rescale factor drawn from the summed kinetic energy leaves the same velocities.
    c = res["coupled"]
    scale = max(abs(c["single"][k]) for k in ("LJ", "COUL_SR", "COUL_RECIP", "COUL_EXCL"))
    for k in ("LJ", "COUL_SR", "COUL_RECIP", "COUL_EXCL", "POT"):
        assert abs(c["dd"][k] - c["single"][k]) < 2e-5 * scale, (k, c)
    assert abs(c["dd"]["KIN"] - c["single"]["KIN"]) < 1e-4 * c["single"]["KIN"], c
    assert abs(c["dd"]["TEMP"] - c["single"]["TEMP"]) < 0.05, c
    assert c["vir_max_diff"] < 2e-4 * c["vir_scale"] and c["pres_max_diff"] < 3.0, c
    assert c["dv_rel_rms"] < 1e-3, c
    assert max(abs(v) for v in c["com_v"]) < 1e-5, c
    # C-rescale across the domains (three coupling steps in 21): every domain draws the single handle's scaling factors
    # from the summed pressure, so the cell ends where the single handle's does and the atoms with it
    p = res["npt"]
    assert p["box_change"] > 1e-7, p                      # the cell did move
    assert p["box_rel_diff"] < 1e-6, p
    assert p["dx_rms"] < 5e-6, p
    assert abs(p["pot_dd"] - p["pot_single"]) < 2e-5 * abs(p["pot_single"]) and p["pres_max_diff"] < 3.0, p
    # Parrinello-Rahman across the domains (the barostat of production.mdp): the same box velocity on every domain
    q = res["npt_pr"]
    assert q["box_change"] > 1e-7 and q["box_rel_diff"] < 1e-6 and q["dx_rms"] < 5e-6, q
    assert abs(q["pot_dd"] - q["pot_single"]) < 2e-5 * abs(q["pot_single"]) and q["pres_max_diff"] < 3.0, q
    # the long-range dispersion correction (DispCorr = EnerPres) is in both: the energy term itself, and through the
    # pressure the cell above
    assert p["dispcorr_single"] < -10.0 and abs(p["dispcorr_dd"] - p["dispcorr_single"]) < 1e-6 * abs(p["dispcorr_single"]), p


@pytest.mark.gpu
def test_device_ordered_collectives_on_one_rank():
    """The RCCL transport orders its reductions on the handle's own streams (no host synchronisation inside a step).
    One rank with every reduction issued for real: thermostat, centre-of-mass removal, C-rescale with the dispersion
    correction, the sum of the charge grid; the run has to reproduce the plain single handle."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dd_ordered_rank.py")], capture_output=True, text=True, timeout=170, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["box_change"] > 1e-7 and r["box_rel_diff"] < 1e-6, r
    assert r["dx_rms"] < 5e-6 and r["dv_rel_rms"] < 1e-3, r
    assert abs(r["pot_dd"] - r["pot_single"]) < 2e-5 * abs(r["pot_single"]) and r["pres_max_diff"] < 3.0, r
    assert max(abs(c) for c in r["com_v"]) < 1e-5, r


@pytest.mark.gpu
def test_decomposed_path_at_the_tiled_million_atom_size():
    """Config 5 at its size on one GPU: the tiled water box (2x2x2 copies of the 12.5 nm dodecahedron, 1.1 M atoms)
    through the decomposed path with one rank against the plain single handle on the same state, plus the properties
    a water box has to keep (rigid molecules, no net momentum gained)."""
    import numpy as np
    import mdx  # noqa: F401
    from mdx import synth, engine as eng, domdec
    base = synth.water_box(d=12.5)
    big = synth.tiled(base, (2, 2, 2))
    assert big.natoms > 1_000_000
    nst = 12
    dd = domdec.DecomposedWater(big, device=0, rc=1.0, dt=0.002, nstlist=10, ref_t=None, rlist=1.1)
    assert dd.world == 1 and dd.n_halo == 0 and dd.n_home == big.natoms // 3
    dd.step(nst); dd.synchronize()
    x, v = dd.gather_state()
    tiles = dd.engine.list_stats()["cluster_pairs"]
    dd.close()
    e = eng.Engine(big, rc=1.0, dt=0.002, nstlist=10, rlist=1.1, nstcalcenergy=0, nstcomm=0)
    e.step(nst); e.synchronize()
    xr, vr, _ = e.get_state()
    tiles_ref = e.list_stats()["cluster_pairs"]
    e.close()
    dx = x - xr
    dx -= np.round(dx @ np.linalg.inv(big.box)) @ big.box
    assert np.sqrt((dx ** 2).mean()) < 2e-6 and np.abs(dx).max() < 1e-4
    assert abs(tiles - tiles_ref) < 0.01 * tiles_ref
    # rigid water geometry and momentum
    x = x.astype(np.float64)
    doh = np.linalg.norm(x[0::3] - x[1::3], axis=1); dhh = np.linalg.norm(x[1::3] - x[2::3], axis=1)
    assert np.abs(doh / 0.09572 - 1).max() < 1e-4 and np.abs(dhh / 0.15139 - 1).max() < 1e-4
    p0 = (big.mass[:, None] * big.v).sum(0); p1 = (big.mass[:, None] * v).sum(0)
    prms = np.sqrt((big.mass[:, None] * v) ** 2).mean() * np.sqrt(big.natoms)
    assert np.abs(p1 - p0).max() < 0.05 * prms            # the mesh part conserves momentum to PME accuracy only
