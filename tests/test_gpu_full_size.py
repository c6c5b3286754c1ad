"""The bench workload at its full size (pseudo-complex in a 12.5 nm rhombic dodecahedron, ~1.4e5 atoms, the settings of
/root/reference/mdp/production.mdp): every force term against the float64 oracle, the exact pair count, and
size-independent properties of the step (momentum, constraint geometry, restart reproducibility)."""
import numpy as np
import pytest

import mdx  # noqa: F401
from mdx import synth, engine as eng
from oracle import mdref as R

pytestmark = pytest.mark.gpu
FTOL, ETOL = 1e-5, 1e-5          # relative RMS of the forces, relative energy: the float32 bar of the small cases


def relrms(a, b):
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


@pytest.fixture(scope="module")
def full():
    s = synth.pseudo_complex(d=12.5)
    s.x = s.x.astype(np.float32).astype(np.float64)
    e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=20, nstcalcenergy=10, tc_groups=["Protein", "Non-Protein"], ref_t=[300.0] * 2,
                   tau_t=[0.1] * 2, nsttcouple=10, dispcorr=2)
    yield s, e
    e.close()


def test_full_size_forces_against_oracle(full):
    s, e = full
    assert s.natoms > 130000
    f, en = e.compute(eng.F_NONBONDED)
    o = R.nonbonded(s.x, s.charge, s.sigma, s.eps, s.box, 1.0, e.beta, s.excl_start, s.excl_list, flags=1, nthreads=8)
    assert relrms(f, o["f"]) < FTOL
    assert abs(en["LJ"] - o["e_lj"]) < ETOL * abs(o["e_lj"])
    assert abs(en["COUL_SR"] - o["e_coul_sr"]) < ETOL * max(abs(o["e_coul_sr"]), abs(o["e_lj"]))
    st = e.list_stats()
    # the count of pairs inside the cutoff is exact up to the pairs whose float32 and float64 r^2 fall on different
    # sides of rc^2 (about 1.5e-7 of them: a handful in 3e7)
    assert abs(st["pairs_within_rc"] - o["npairs"]) <= 64
    f, en = e.compute(eng.F_PME)
    Ep, Fp, _ = R.pme_recip(s.x, s.charge, s.box, e.beta, e.grid)
    assert relrms(f, Fp) < FTOL
    assert abs(en["COUL_RECIP"] - Ep) < ETOL * abs(Ep)
    f, en = e.compute(eng.F_BONDED | eng.F_EXCL)
    ox = R.nonbonded(s.x, s.charge, s.sigma, s.eps, s.box, 1.0, e.beta, s.excl_start, s.excl_list, flags=2, nthreads=8)
    fb = ox["f"] + R.bonds(s.x, s.bonds, s.bond_b0, s.bond_kb)[1] + R.angles(s.x, s.angles, s.angle_th0, s.angle_k)[1] \
        + R.dihedrals(s.x, s.dihedrals, s.dih_phi0, s.dih_k, s.dih_mult)[1] \
        + R.pairs14(s.x, s.pairs14, s.pair_c6, s.pair_c12, s.pair_qq)[2]
    assert relrms(f, fb) < FTOL
    assert abs(en["COUL_EXCL"] - ox["e_excl"]) < ETOL * abs(ox["e_excl"])


def test_full_size_real_space_forces_sum_to_zero(full):
    s, e = full
    f, _ = e.compute(eng.F_NONBONDED | eng.F_BONDED | eng.F_EXCL)
    f = f.astype(np.float64)
    frms = np.sqrt((f ** 2).sum(1).mean())
    assert np.abs(f.sum(0)).max() < 1e-5 * frms * np.sqrt(s.natoms)


def test_full_size_steps_keep_constraints_and_momentum_and_repeat(full):
    s, e = full
    e.set_state(s.x, s.v, s.box); e.set_step(0)
    e.step(40)
    x1, v1, _ = e.get_state()
    x1 = x1.astype(np.float64)
    # rigid water and h-bond constraints hold after 40 steps with one list rebuild
    w = s.settles
    d = np.linalg.norm(x1[w[:, 0]] - x1[w[:, 1]], axis=1)
    assert np.abs(d / s.settle_doh - 1).max() < 3e-5
    c = s.constraints
    d = np.linalg.norm(x1[c[:, 0]] - x1[c[:, 1]], axis=1)
    assert np.abs(d / s.constraint_len - 1).max() < 2e-4
    # the centre-of-mass velocity stays three orders below a thermal velocity (it is removed every `nstcomm` = 100
    # steps; in between only the mesh part of the force has a net sum, of PME accuracy)
    p = (s.mass[:, None] * v1.astype(np.float64)).sum(0) / s.mass.sum()
    assert np.abs(p).max() < 1e-3
    assert np.isfinite(x1).all() and np.abs(x1 - s.x).max() < 0.5
    # the same 40 steps from the same state give the same trajectory up to the order of the float32 force sums
    e.set_state(s.x, s.v, s.box); e.set_step(0)
    e.step(40)
    x2, _, _ = e.get_state()
    assert np.sqrt(((x2 - x1) ** 2).sum(1).mean()) < 2e-5
