"""The CPU baseline that bench.py times (oracle/cpu_md.*: single precision, OpenMP, Verlet list) is itself checked
against the float64 oracle, so the number reported beside the GPU figure comes from a computation known to be right."""
import numpy as np

import mdx  # noqa: F401
from mdx import synth, units as U
from oracle import mdref as R, cpu_md


def test_cpu_baseline_forces_and_step_match_the_oracle():
    s = synth.pseudo_complex(d=6.0, protein_radius=1.0, n_active=60, ligand_offset=1.2)
    s.x = s.x.astype(np.float32).astype(np.float64)
    beta = U.ewald_beta(1.0); grid = U.pme_grid(s.box, 0.16)
    c = cpu_md.CpuMD(s, 1.0, beta, grid, 0.002, rlist=1.1, nstlist=10, threads=2)
    f = c.forces(want_energy=True).astype(np.float64)
    ref = R.all_forces(s, 1.0, beta, grid, nthreads=4)
    assert np.sqrt(((f - ref["f"]) ** 2).sum() / (ref["f"] ** 2).sum()) < 2e-5
    nb = ref["nb"]
    assert c.npairs == nb["npairs"] and c.npairs_listed > c.npairs
    assert abs(c.energies["lj"] - nb["e_lj"]) < 1e-5 * abs(nb["e_lj"])
    assert abs(c.energies["coul_sr"] - nb["e_coul_sr"]) < 1e-5 * abs(nb["e_lj"])
    assert abs(c.energies["excl"] - nb["e_excl"]) < 1e-5 * abs(nb["e_excl"])
    assert abs(c.energies["recip"] - ref["recip"][0]) < 1e-5 * abs(ref["recip"][0])
    v0 = s.v.astype(np.float32).astype(np.float64)
    xr, vr, _ = R.md_step(s, v0, 1.0, beta, grid, 0.002, nthreads=4)
    c.step(1)
    assert np.abs(c.x - xr).max() < 5e-6 and np.abs(c.v - vr).max() < 3e-3
    # the list is reused: after a few steps the forces from the old list still match a fresh evaluation
    c.step(5)
    f_old_list = c.forces().astype(np.float64).copy()
    t = synth.pseudo_complex(d=6.0, protein_radius=1.0, n_active=60, ligand_offset=1.2)
    t.x = c.x.astype(np.float64)
    fresh = cpu_md.CpuMD(t, 1.0, beta, grid, 0.002, rlist=1.1, nstlist=10, threads=2)
    f_new = fresh.forces().astype(np.float64)
    assert np.sqrt(((f_old_list - f_new) ** 2).sum() / (f_new ** 2).sum()) < 2e-5
    c.close(); fresh.close()
