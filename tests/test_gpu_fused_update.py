"""The fused update (leap-frog + SETTLE + LINCS in one launch, which also leaves the next step's cluster-ordered
coordinates so that the step needs no prepare launch) against the float64 oracle and against the separate kernels."""
import numpy as np
import pytest

import mdx  # noqa: F401
from mdx import engine as eng
from oracle import mdref as R

pytestmark = pytest.mark.gpu


def _copy_with(s, x, v):
    import copy
    t = copy.copy(s)
    t.x = np.asarray(x, dtype=np.float64); t.v = np.asarray(v, dtype=np.float64)
    return t


def test_two_steps_in_one_call_match_oracle(small_complex):
    """Two steps of one mdx_step call: the second runs on the sorted coordinates the first step's update wrote."""
    s = small_complex
    v0 = s.v.astype(np.float32).astype(np.float64)
    e = eng.Engine(s, rc=1.0, nstcalcenergy=0, nstcomm=0, dt=0.002, lincs_order=4, lincs_iter=1)
    e.set_state(s.x, v0, s.box)
    e.step(2)
    x2, v2, _ = e.get_state()
    e.close()
    xr, vr, _ = R.md_step(_copy_with(s, s.x, v0), v0, 1.0, e.beta, e.grid, 0.002, nthreads=8)
    xr, vr, _ = R.md_step(_copy_with(s, xr, vr), vr, 1.0, e.beta, e.grid, 0.002, nthreads=8)
    assert np.abs(x2 - xr).max() < 8e-6           # twice the one-step bound of test_gpu_dynamics
    assert np.abs(v2 - vr).max() < 4e-3


@pytest.mark.parametrize("pcoupl", [0, 2])
def test_fused_and_separate_kernels_agree(small_complex, pcoupl):
    """25 steps across thermostat steps, energy/pressure-coupling steps (separate kernels either way) and in calls of
    uneven length: the same trajectory with the fused update on and off, up to the order of the float32 force sums."""
    s = small_complex
    out = []
    for fused in (1, 0):
        e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=10, nstcalcenergy=10, tc_groups=["Protein", "Non-Protein"], ref_t=[300.0] * 2,
                       tau_t=[0.1] * 2, nsttcouple=5, pcoupl=pcoupl, tau_p=2.0, ref_p=1.0, compressibility=4.5e-5, nstpcouple=10,
                       dispcorr=2, seed=7)
        e.debug_set(12, fused)
        e.set_state(s.x, s.v, s.box); e.set_step(0)
        for n in (1, 7, 3, 14):
            e.step(n)
        x, v, box = e.get_state()
        fr = e.pop_energy_frames()
        out.append((x.astype(np.float64), v.astype(np.float64), box, fr))
        e.close()
    (xa, va, ba, fa), (xb, vb, bb, fb) = out
    assert np.sqrt(((xa - xb) ** 2).sum(1).mean()) < 5e-6
    assert np.sqrt(((va - vb) ** 2).sum(1).mean()) < 5e-3 * np.sqrt((vb ** 2).sum(1).mean())
    assert np.allclose(ba, bb, rtol=1e-6, atol=0)
    assert len(fa) == len(fb) == 3
    pot = eng.E["POT"]
    assert np.allclose(fa[:, 2 + pot], fb[:, 2 + pot], rtol=2e-6)


def test_bonded_terms_in_the_spreading_launch_agree_with_their_own_launch(small_complex):
    """On steps without energies the bonded terms ride in the charge-spreading launch and the update waits for the PME
    stream only (thermostat steps and frames put work on the bonded stream again). The same 23 steps with the bonded
    launch kept separate (mdx_debug_set key 2, bit 0x4000000), up to the order of the float32 force sums; a trajectory
    frame captured on the way arrives either way."""
    s = small_complex
    out = []
    for separate in (0, 1):
        e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=10, nstcalcenergy=10, tc_groups=["Protein", "Non-Protein"], ref_t=[300.0] * 2,
                       tau_t=[0.1] * 2, nsttcouple=5, seed=7)
        if separate:
            e.debug_set(2, 0x4000000)
        e.frame_ring(2)
        e.set_state(s.x, s.v, s.box); e.set_step(0)
        e.step(12)
        e.capture_frame(0)
        for n in (1, 7, 3):
            e.step(n)
        x, v, _ = e.get_state()
        _, xf, _, _ = e.frame_wait(0)
        out.append((x.astype(np.float64), v.astype(np.float64), xf.astype(np.float64)))
        e.close()
    (xa, va, fa), (xb, vb, fb) = out
    assert np.sqrt(((xa - xb) ** 2).sum(1).mean()) < 5e-6
    assert np.sqrt(((va - vb) ** 2).sum(1).mean()) < 5e-3 * np.sqrt((vb ** 2).sum(1).mean())
    assert np.sqrt(((fa - fb) ** 2).sum(1).mean()) < 5e-6


def test_fused_update_water_keeps_geometry_over_a_search(small_water):
    s = small_water
    e = eng.Engine(s, rc=1.0, nstcalcenergy=0, nstlist=10, dt=0.002)
    e.step(35)
    x, v, _ = e.get_state()
    e.close()
    x = x.astype(np.float64)
    d = np.linalg.norm(x[0::3] - x[1::3], axis=1)
    assert np.abs(d / 0.09572 - 1).max() < 3e-5
    d = np.linalg.norm(x[1::3] - x[2::3], axis=1)
    assert np.abs(d / 0.15139 - 1).max() < 3e-5
    assert np.isfinite(v).all()


def _with_chain_constraints(s, nchain=5):
    """A copy of the system in which `nchain` consecutive heavy-atom bonds of the protein are constrained as well (at
    their present lengths), so that the h-bond components along them merge into one component of more than four
    constraints (at most 16: the limit of the LINCS kernels)."""
    import copy
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    c = np.asarray(s.constraints)
    in_c = np.zeros(s.natoms, bool); in_c[c.ravel()] = True
    have = {tuple(sorted(p)) for p in c.tolist()}
    water = np.zeros(s.natoms, bool); water[np.asarray(s.settles).ravel()] = True
    # walk along the bonded chain from the first heavy atom that carries constrained hydrogens
    nb = {}
    for a, b in np.asarray(s.bonds).tolist():
        if tuple(sorted((a, b))) in have or water[a] or water[b]:
            continue
        nb.setdefault(a, []).append(b); nb.setdefault(b, []).append(a)
    start = next(a for a in sorted(nb) if in_c[a])
    chain, seen, cur = [], {start}, start
    while len(chain) < nchain:
        nxt = next((b for b in nb.get(cur, []) if b not in seen), None)
        if nxt is None:
            break
        chain.append((cur, nxt)); seen.add(nxt); cur = nxt
    assert len(chain) == nchain
    x = np.asarray(s.x, dtype=np.float64)
    extra = np.array(chain, dtype=c.dtype)
    t = copy.copy(s)
    t.constraints = np.concatenate([c, extra])
    t.constraint_len = np.concatenate([np.asarray(s.constraint_len), np.linalg.norm(x[extra[:, 0]] - x[extra[:, 1]], axis=1)]).astype(
        np.asarray(s.constraint_len).dtype)
    cc = t.constraints
    g = coo_matrix((np.ones(len(cc)), (cc[:, 0], cc[:, 1])), shape=(s.natoms, s.natoms))
    _, lab = connected_components(g, directed=False)
    sizes = np.bincount(lab[cc[:, 0]])
    return t, int(sizes.max())


def test_large_constraint_component(small_complex):
    """A component of 5 to 16 coupled constraints: one step against the oracle's LINCS of the same order, and the fused
    update (component on 17-atom LDS columns) against the separate LINCS kernel over several steps."""
    s, kmax = _with_chain_constraints(small_complex, nchain=3)          # 12 coupled constraints
    assert 4 < kmax <= 16
    v0 = s.v.astype(np.float32).astype(np.float64)
    res = []
    for fused in (1, 0):
        e = eng.Engine(s, rc=1.0, nstcalcenergy=0, nstcomm=0, dt=0.002, lincs_order=4, lincs_iter=1)
        e.debug_set(12, fused)
        e.set_state(s.x, v0, s.box)
        e.step(1)
        x1, v1, _ = e.get_state()
        e.step(6)
        x7, _, _ = e.get_state()
        res.append((x1.astype(np.float64), v1.astype(np.float64), x7.astype(np.float64)))
        e.close()
    xr, vr, _ = R.md_step(_copy_with(s, s.x, v0), v0, 1.0, e.beta, e.grid, 0.002, nthreads=8)
    for x1, v1, _ in res:
        assert np.abs(x1 - xr).max() < 4e-6
        assert np.abs(v1 - vr).max() < 2e-3
    assert np.abs(res[0][0] - res[1][0]).max() < 1e-6
    assert np.sqrt(((res[0][2] - res[1][2]) ** 2).sum(1).mean()) < 2e-6
    c = s.constraints
    d = np.linalg.norm(res[0][2][c[:, 0]] - res[0][2][c[:, 1]], axis=1)
    assert np.abs(d / s.constraint_len - 1).max() < 2e-4
