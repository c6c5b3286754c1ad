"""GPU parity: the HIP path through the C ABI against the float64 oracle on the same float32
inputs. Tolerances follow the north-star bar: per-atom force relative RMS <= 1e-5, energies
within 1e-5 relative (mixed precision)."""
import numpy as np
import pytest

import mdx  # noqa: F401
from mdx import synth, engine as eng
from oracle import mdref as R

pytestmark = pytest.mark.gpu
FTOL = 1e-5
ETOL = 1e-5


def relrms(a, b):
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


@pytest.fixture(scope="module")
def eng_water(small_water):
    e = eng.Engine(small_water, rc=1.0, nstcalcenergy=1)
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng_complex(small_complex):
    e = eng.Engine(small_complex, rc=1.0, nstcalcenergy=1, use_posres=True)
    yield e
    e.close()


@pytest.mark.parametrize("which", ["water", "complex"])
def test_nonbonded_forces_energies_virial(which, eng_water, eng_complex, small_water, small_complex):
    e, s = (eng_water, small_water) if which == "water" else (eng_complex, small_complex)
    f, en = e.compute(eng.F_NONBONDED)
    o = R.nonbonded(s.x, s.charge, s.sigma, s.eps, s.box, 1.0, e.beta, s.excl_start, s.excl_list, flags=1, nthreads=8)
    assert relrms(f, o["f"]) < FTOL
    assert abs(en["LJ"] - o["e_lj"]) < ETOL * abs(o["e_lj"])
    assert abs(en["COUL_SR"] - o["e_coul_sr"]) < ETOL * max(abs(o["e_coul_sr"]), abs(o["e_lj"]))
    v, _ = e.virial()
    assert np.abs(v - o["vir"]).max() < 1e-4 * np.abs(o["vir"]).max()
    # pair count from the kernel equals the oracle's exact count
    st = e.list_stats()
    # (a pair whose float32 and float64 r^2 fall on different sides of rc^2 may differ: none or a couple at this size)
    assert abs(st["pairs_within_rc"] - o["npairs"]) <= 2
    assert st["cluster_pairs"] * 64 >= st["pairs_within_rc"]


@pytest.mark.parametrize("which", ["water", "complex"])
def test_pme_same_grid(which, eng_water, eng_complex, small_water, small_complex):
    e, s = (eng_water, small_water) if which == "water" else (eng_complex, small_complex)
    f, en = e.compute(eng.F_PME)
    Ep, Fp, vp = R.pme_recip(s.x, s.charge, s.box, e.beta, e.grid)
    assert relrms(f, Fp) < FTOL
    assert abs(en["COUL_RECIP"] - Ep) < ETOL * abs(Ep)
    v, _ = e.virial()
    assert np.abs(v - vp).max() < 1e-4 * np.abs(vp).max()


@pytest.mark.parametrize("which", ["water", "complex"])
def test_exclusion_correction(which, eng_water, eng_complex, small_water, small_complex):
    e, s = (eng_water, small_water) if which == "water" else (eng_complex, small_complex)
    f, en = e.compute(eng.F_EXCL)
    o = R.nonbonded(s.x, s.charge, s.sigma, s.eps, s.box, 1.0, e.beta, s.excl_start, s.excl_list, flags=2)
    assert relrms(f, o["f"]) < FTOL
    # The term is the difference of two large sums (pair corrections against the Ewald self energy, 3.3e5 kJ/mol each in
    # the water box for a total of -3.9e3): float32 pair sums resolve it to a few 1e-7 of those, which is 1e-5 of the total.
    self_term = 138.935458 * e.beta / np.sqrt(np.pi) * float((s.charge ** 2).sum())
    assert abs(en["COUL_EXCL"] - o["e_excl"]) < ETOL * abs(o["e_excl"]) + 3e-7 * self_term


def test_bonded_terms(eng_complex, small_complex):
    e, s = eng_complex, small_complex
    # thermal-like distortion so that every term carries a real force
    rng = np.random.default_rng(7)
    x = (s.x + rng.normal(scale=0.004, size=s.x.shape)).astype(np.float32).astype(np.float64)
    e.set_state(x, None, s.box)
    f, en = e.compute(eng.F_BONDED | eng.F_POSRES)
    eb, fb, _ = R.bonds(x, s.bonds, s.bond_b0, s.bond_kb)
    ea, fa, _ = R.angles(x, s.angles, s.angle_th0, s.angle_k)
    ed, fd, _ = R.dihedrals(x, s.dihedrals, s.dih_phi0, s.dih_k, s.dih_mult)
    el, ec, fp, _ = R.pairs14(x, s.pairs14, s.pair_c6, s.pair_c12, s.pair_qq)
    epr, fpr, _ = R.posres(x, s.box, s.posres_idx, s.posres_ref.astype(np.float32).astype(np.float64), s.posres_k)
    assert relrms(f, fb + fa + fd + fp + fpr) < FTOL
    for name, ref in (("BOND", eb), ("ANGLE", ea), ("DIH", ed), ("LJ14", el), ("COUL14", ec), ("POSRES", epr)):
        assert abs(en[name] - ref) < ETOL * max(abs(ref), 1.0), name
    e.set_state(s.x, s.v, s.box)


def test_bonded_terms_across_distant_atom_indices(small_complex):
    """Bonds, angles and dihedrals whose atoms lie thousands of indices apart (cross-links, interactions listed out of
    atom order): the bonded kernel sums a workgroup's forces in an LDS window over 1024 atom indices from the lowest one
    it sees and adds everything outside that window to the force array directly; both paths against the oracle."""
    import copy
    s = copy.deepcopy(small_complex)
    n = s.natoms
    assert n > 5000
    rng = np.random.default_rng(11)
    lo = rng.integers(0, 200, size=40); hi = n - 1 - rng.integers(0, 200, size=40)
    mid = n // 2 + rng.integers(-100, 100, size=40)
    s.bonds = np.concatenate([s.bonds, np.stack([lo, hi], 1).astype(np.int32)])
    s.bond_b0 = np.concatenate([s.bond_b0, np.full(40, 2.5)]); s.bond_kb = np.concatenate([s.bond_kb, np.full(40, 5.0)])
    s.angles = np.concatenate([s.angles, np.stack([hi, lo, mid], 1).astype(np.int32)])
    s.angle_th0 = np.concatenate([s.angle_th0, np.full(40, 1.9)]); s.angle_k = np.concatenate([s.angle_k, np.full(40, 1.0)])
    s.dihedrals = np.concatenate([s.dihedrals, np.stack([hi, mid, lo, lo + 300], 1).astype(np.int32)])
    s.dih_phi0 = np.concatenate([s.dih_phi0, np.full(40, 0.3)]); s.dih_k = np.concatenate([s.dih_k, np.full(40, 0.5)])
    s.dih_mult = np.concatenate([s.dih_mult, np.full(40, 2, dtype=s.dih_mult.dtype)])
    e = eng.Engine(s, rc=1.0, nstcalcenergy=1, use_posres=True)
    try:
        # thermal-like distortion as in test_bonded_terms (at the undistorted geometry the bonded forces nearly cancel and
        # float32 trigonometry shows at 2e-5 of what is left)
        x = (s.x + np.random.default_rng(7).normal(scale=0.004, size=s.x.shape)).astype(np.float32).astype(np.float64)
        e.set_state(x, None, s.box)
        f, en = e.compute(eng.F_BONDED)
        eb, fb, _ = R.bonds(x, s.bonds, s.bond_b0, s.bond_kb)
        ea, fa, _ = R.angles(x, s.angles, s.angle_th0, s.angle_k)
        ed, fd, _ = R.dihedrals(x, s.dihedrals, s.dih_phi0, s.dih_k, s.dih_mult)
        el, ec, fp, _ = R.pairs14(x, s.pairs14, s.pair_c6, s.pair_c12, s.pair_qq)
        ref = fb + fa + fd + fp
        assert relrms(f, ref) < FTOL
        for name, ee in (("BOND", eb), ("ANGLE", ea), ("DIH", ed)):
            assert abs(en[name] - ee) < ETOL * max(abs(ee), 1.0), name
        # the far ends (water atoms at the top of the index range, which carry nothing but these cross-links) did receive
        # their forces: float32 trigonometry on nanometre-long arms limits them to 1e-4
        far = np.unique(hi)
        assert np.abs(ref[far]).max() > 0
        assert relrms(f[far], ref[far]) < 1e-4
    finally:
        e.close()


def test_total_force_and_newton(eng_complex, small_complex):
    e, s = eng_complex, small_complex
    f, en = e.compute(eng.F_ALL & ~eng.F_PULL & ~eng.F_POSRES)
    ref = R.all_forces(s, 1.0, e.beta, e.grid, nthreads=8)
    assert relrms(f, ref["f"]) < FTOL
    frms = np.sqrt((f.astype(np.float64) ** 2).sum(1).mean())
    # real-space + bonded parts obey Newton's third law; the mesh part only to PME accuracy
    f2, _ = e.compute(eng.F_NONBONDED | eng.F_BONDED | eng.F_EXCL)
    assert np.abs(f2.astype(np.float64).sum(0)).max() < 1e-4 * frms * np.sqrt(s.natoms)


def test_translation_and_image_invariance(eng_water, small_water):
    e, s = eng_water, small_water
    _, en0 = e.compute(eng.F_NONBONDED | eng.F_EXCL)
    f0, _ = e.compute(eng.F_NONBONDED | eng.F_EXCL)
    # rigid translation (real-space terms are exactly translation invariant up to rounding)
    e.set_state(s.x + np.array([0.3713, -0.211, 0.157]), None, s.box)
    f1, en1 = e.compute(eng.F_NONBONDED | eng.F_EXCL)
    for k in ("LJ", "COUL_SR", "COUL_EXCL"):
        assert abs(en1[k] - en0[k]) < 2e-5 * abs(en0[k])
    assert relrms(f1, f0) < 5e-5
    # whole molecules moved to other periodic images: identical physics
    rng = np.random.default_rng(8)
    sh = rng.integers(-1, 2, size=(s.natoms // 3, 3)) @ s.box
    e.set_state(s.x + np.repeat(sh, 3, axis=0), None, s.box)
    f2, en2 = e.compute(eng.F_NONBONDED | eng.F_EXCL | eng.F_PME)
    e.set_state(s.x, s.v, s.box)
    f3, en3 = e.compute(eng.F_NONBONDED | eng.F_EXCL | eng.F_PME)
    for k in ("LJ", "COUL_SR", "COUL_RECIP"):
        assert abs(en2[k] - en3[k]) < 2e-5 * abs(en3[k])
    assert relrms(f2, f3) < 5e-5


def test_pair_list_covers_every_pair_within_cutoff(eng_water, small_water):
    """Energy from the cluster list equals the oracle's exact all-pairs sum, so no pair within
    rc is missing and none is counted twice; rlist buffer does not change the result."""
    s = small_water
    e2 = eng.Engine(s, rc=1.0, rlist=1.18, nstcalcenergy=1)
    _, en_a = eng_water.compute(eng.F_NONBONDED)
    _, en_b = e2.compute(eng.F_NONBONDED)
    e2.close()
    assert abs(en_a["LJ"] - en_b["LJ"]) < 1e-6 * abs(en_a["LJ"])
    assert abs(en_a["COUL_SR"] - en_b["COUL_SR"]) < 1e-5 * abs(en_a["LJ"])


def test_cutoff_1p2_pull_settings(small_water):
    """rc = 1.2 nm as /root/reference/mdp/pull_short.mdp:19-20 sets it."""
    s = synth.water_box(d=4.2)
    s.x = s.x.astype(np.float32).astype(np.float64)
    e = eng.Engine(s, rc=1.2, nstcalcenergy=1)
    f, en = e.compute(eng.F_NONBONDED)
    o = R.nonbonded(s.x, s.charge, s.sigma, s.eps, s.box, 1.2, e.beta, s.excl_start, s.excl_list, flags=1, nthreads=8)
    e.close()
    assert relrms(f, o["f"]) < FTOL
    assert abs(en["LJ"] - o["e_lj"]) < ETOL * abs(o["e_lj"])


def test_pipeline_dialect_system_total_force():
    """The system parsed from the pipeline-style topology fixture (pdb2gmx-like protein block,
    acpype-like ligand include, tip3p/ions includes) evaluated by the HIP path and by the oracle."""
    import os
    from mdx.io import gro, top
    td = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "topdialect")
    g = gro.read_gro(os.path.join(td, "complex.gro"))
    s = top.build_system(top.read_top(os.path.join(td, "ion.top"), defines={"POSRES": ""}), g, constraints="h-bonds", posres_ref=g)
    s.x = s.x.astype(np.float32).astype(np.float64)
    # 2.4 nm cube: the cutoff and pair-list radius have to fit twice into the box
    e = eng.Engine(s, rc=0.9, rlist=1.0, nstcalcenergy=1, use_posres=True)
    f, en = e.compute(eng.F_ALL & ~eng.F_PULL & ~eng.F_POSRES)
    ref = R.all_forces(s, 0.9, e.beta, e.grid, nthreads=8)
    e.close()
    assert relrms(f, ref["f"]) < FTOL
    assert np.isfinite(list(en.values())).all()
