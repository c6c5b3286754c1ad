"""COM pull (`pull = yes`, /root/reference/mdp/pull_short.mdp:36-46) through the C ABI against the float64 oracle:
per-atom forces, scalar force, energy, virial, the work sums of the pull records, and groups that straddle a face of
the rhombic dodecahedron."""
import copy

import numpy as np
import pytest

import mdx  # noqa: F401
from mdx import engine as eng
from oracle import mdref as R

pytestmark = pytest.mark.gpu

FORCE_TOL = 1e-5       # relative RMS of the per-atom forces, the bar of the other force terms


def _relrms(a, b):
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


def _engine(s, **kw):
    return eng.Engine(s, rc=1.0, dt=0.001, nstcalcenergy=100, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1], **kw)


def test_constant_force_per_atom_forces(small_complex):
    """k = 500 as in /root/reference/mdp/pull_short.mdp:46: every atom of ActiveSite and MOL carries its
    mass-weighted share of -/+ f rhat; no other atom carries anything."""
    s = small_complex
    g1, g2 = s.groups["ActiveSite"], s.groups["MOL"]
    e = _engine(s)
    e.set_pull(g1, g2, type="constant-force", k=500.0, nstout=50)
    f, en = e.compute(eng.F_PULL)
    vir, _ = e.virial()
    e.close()
    o = R.pull_forces(s.x, s.box, s.mass, g1, g2, type="constant-force", k=500.0)
    assert o["f"] == -500.0
    assert _relrms(f.astype(np.float64), o["forces"]) < FORCE_TOL
    others = np.setdiff1d(np.arange(s.natoms), np.concatenate([g1, g2]))
    assert np.abs(f[others]).max() == 0.0
    # single atoms, not only the RMS: the heaviest and the lightest atom of each group
    for g in (g1, g2):
        for a in (g[np.argmax(s.mass[g])], g[np.argmin(s.mass[g])]):
            assert np.abs(f[a] - o["forces"][a]).max() < 2e-6 * np.abs(o["forces"][a]).max() + 1e-9
    assert abs(f.astype(np.float64).sum(0)).max() < 1e-3           # action = reaction over the two groups
    assert abs(en["PULL"] - o["epot"]) < 1e-6 * abs(o["epot"])
    assert np.abs(vir - o["virial"]).max() < 1e-5 * np.abs(o["virial"]).max()


def test_pull_virial_enters_the_pressure(small_complex):
    """The pull virial is part of the virial the pressure is formed from: pressure with the pull term minus pressure
    without it = -2/(3V) tr(Xi_pull) * PRESFAC on the diagonal average."""
    s = small_complex
    g1, g2 = s.groups["ActiveSite"], s.groups["MOL"]
    e = _engine(s)
    e.set_pull(g1, g2, type="constant-force", k=500.0, nstout=50)
    terms = eng.F_BONDED
    e.compute(terms)
    v0, p0 = e.virial()
    e.compute(terms | eng.F_PULL)
    v1, p1 = e.virial()
    e.close()
    o = R.pull_forces(s.x, s.box, s.mass, g1, g2, type="constant-force", k=500.0)
    assert np.abs((v1 - v0) - o["virial"]).max() < 1e-4 * np.abs(o["virial"]).max()
    vol = abs(np.linalg.det(s.box))
    dp = -2.0 / vol * o["virial"] * R.PRESFAC
    assert np.abs((p1 - p0) - dp).max() < 1e-4 * np.abs(dp).max() + 1e-6


def test_umbrella_rate_force_and_work_over_200_steps(small_complex):
    """Umbrella with a moving reference (k = 1000, rate = -0.001 nm/ps of /root/reference/config.yaml:38-39 scaled
    up a hundredfold for signal): at each of 200 consecutive steps the device's (value, force, work) record against the
    oracle evaluated on the device's own trajectory."""
    s = small_complex
    g1, g2 = s.groups["ActiveSite"], s.groups["MOL"]
    k, rate, dt = 1000.0, -0.1, 0.001
    e = _engine(s)
    e.set_pull(g1, g2, type="umbrella", k=k, rate=rate, nstout=1)
    xs = []
    for _ in range(200):
        x, _, box = e.get_state()
        xs.append(x.astype(np.float64))
        e.step(1)
    rec = e.pop_pull_records()
    e.close()
    assert len(rec) == 200
    r0 = R.pull_forces(xs[0], s.box, s.mass, g1, g2, type="umbrella", k=k)["r"]
    ro, fo = [], []
    for n, x in enumerate(xs):
        o = R.pull_forces(x, s.box, s.mass, g1, g2, type="umbrella", k=k, rate=rate, t=n * dt, init=r0)
        ro.append(o["r"]); fo.append(o["f"])
    ro, fo = np.array(ro), np.array(fo)
    assert np.allclose(rec[:, 0], np.arange(200) * dt, atol=1e-12)
    assert np.abs(rec[:, 1] - ro).max() < 2e-6                       # float32 coordinates
    assert np.abs(rec[:, 2] - fo).max() < k * 4e-6                   # the force follows the value
    wo = R.pull_work("umbrella", k, rate, dt, ro, fo)
    assert np.abs(rec[:, 3] - wo).max() < 1e-5 * max(1e-3, np.abs(wo).max()) + 200 * k * 4e-6 * abs(rate) * dt
    assert abs(fo[-1]) > 10.0                                         # the trap did pull away from the COM distance


def test_umbrella_per_atom_forces_at_a_later_time(small_complex):
    """Umbrella forces on single atoms with the reference displaced: the handle is put at step 400 (t = 0.4 ps) so
    that r_ref = r(0) + rate t differs from r by 0.04 nm."""
    s = small_complex
    g1, g2 = s.groups["ActiveSite"], s.groups["MOL"]
    k, rate = 1000.0, -0.1
    e = _engine(s)
    e.set_pull(g1, g2, type="umbrella", k=k, rate=rate, nstout=50)
    e.compute(eng.F_PULL)                     # fixes the reference value at the starting configuration (start = yes)
    e.set_step(400)
    f, en = e.compute(eng.F_PULL)
    e.close()
    r0 = R.pull_forces(s.x, s.box, s.mass, g1, g2, type="umbrella", k=k)["r"]
    o = R.pull_forces(s.x, s.box, s.mass, g1, g2, type="umbrella", k=k, rate=rate, t=0.4, init=r0)
    assert abs(o["f"] - (-k * 0.04)) < 1e-9
    assert _relrms(f.astype(np.float64), o["forces"]) < 3e-5         # f = -k (r - r_ref): r in float32 at ~1 nm
    assert abs(en["PULL"] - o["epot"]) < 1e-4 * o["epot"]


@pytest.mark.parametrize("face", [0, 1, 2])
def test_groups_straddling_a_dodecahedron_face(small_complex, face):
    """Both groups moved so that the face `face` of the rhombic dodecahedron cuts through them, every atom then put
    back into the cell on its own (molecules broken across the face): the device's COM distance and per-atom forces
    must be those of the intact configuration (the PBC reference atom path of the COM sum)."""
    s = small_complex
    g1, g2 = s.groups["ActiveSite"], s.groups["MOL"]
    o_intact = R.pull_forces(s.x, s.box, s.mass, g1, g2, type="constant-force", k=500.0)
    box = np.asarray(s.box, dtype=np.float64)
    inv = np.linalg.inv(box)
    t = copy.copy(s)
    # move the middle between the two COMs onto the face s_face = 0 (== 1) of the cell
    mid = 0.5 * (o_intact["com1"] + o_intact["com2"])
    frac_mid = mid @ inv
    shift = np.zeros(3); shift[face] = -frac_mid[face]
    x = s.x + (shift @ box)[None, :]
    fr = x @ inv
    both = np.concatenate([g1, g2])
    fr[both] -= np.floor(fr[both])            # per-atom wrap of the pulled atoms only
    x = (fr @ box).astype(np.float32).astype(np.float64)
    side = np.floor((s.x + (shift @ box)[None, :])[both] @ inv)[:, face]
    assert len(np.unique(side)) > 1, "the face does not cut the groups"
    t.x = x
    e = _engine(t)
    e.set_pull(g1, g2, type="constant-force", k=500.0, nstout=50)
    f, _ = e.compute(eng.F_PULL)
    e.step(1)
    rec = e.pop_pull_records()
    e.close()
    o = R.pull_forces(x, s.box, s.mass, g1, g2, type="constant-force", k=500.0)
    assert abs(o["r"] - o_intact["r"]) < 1e-6                       # the oracle itself is image invariant
    assert abs(rec[0, 1] - o_intact["r"]) < 3e-6
    assert _relrms(f.astype(np.float64), o_intact["forces"]) < FORCE_TOL


def test_pull_in_the_spreading_launch_agrees_with_its_own_launch(small_complex):
    """On steps without energies the pull rides in the charge-spreading launch (with the bonded terms); the same 30
    steps with both kept in launches of their own on the bonded stream (mdx_debug_set key 2, bit 0x4000000): same
    records, same trajectory up to the order of the float32 force sums, and the pull did move the ligand (a run
    without the pull ends elsewhere)."""
    s = small_complex
    g1, g2 = s.groups["ActiveSite"], s.groups["MOL"]
    out = []
    for mode in ("merged", "separate", "no pull"):
        e = _engine(s, seed=5)
        if mode != "no pull":
            e.set_pull(g1, g2, type="constant-force", k=5000.0, nstout=1)
        if mode == "separate":
            e.debug_set(2, 0x4000000)
        for n in (1, 12, 17):
            e.step(n)
        x, _, _ = e.get_state()
        rec = e.pop_pull_records() if mode != "no pull" else None
        e.close()
        out.append((x.astype(np.float64), rec))
    (xa, ra), (xb, rb), (xc, _) = out
    assert len(ra) == len(rb) == 30
    assert np.abs(ra[:, 1] - rb[:, 1]).max() < 2e-6 and np.abs(ra[:, 3] - rb[:, 3]).max() < 1e-4 * np.abs(rb[:, 3]).max() + 1e-6
    assert np.sqrt(((xa - xb) ** 2).sum(1).mean()) < 5e-6
    assert np.abs(xa[g2] - xc[g2]).max() > 1e-4
