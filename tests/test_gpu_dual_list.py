"""Dual pair list with a rolling prune (`nstlist = 200` under `verlet-buffer-tolerance`,
/root/reference/mdp/npt_complex.mdp:29-32) through the C ABI: the nonbonded kernel on a pruned list must see exactly
the pairs inside the cutoff that the oracle finds at the same positions, and a run on the dual list must follow the
run on a single list."""
import numpy as np
import pytest

import mdx  # noqa: F401
from mdx import synth, engine as eng
from oracle import mdref as R

pytestmark = pytest.mark.gpu


def _pruned_step_against_oracle(s, nstlist, nstprune, nsteps, rlist=None, rlist_inner=None, exact=True):
    """Steps until the list in use is a pruned one of some age, then one energy step there against the oracle."""
    e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=nstlist, nstprune=nstprune, nstcalcenergy=1, nstcomm=0, rlist=rlist,
                   rlist_inner=rlist_inner)
    try:
        assert e.nstprune == nstprune and e.rc <= e.rlist_inner <= e.rlist
        e.debug_set(11, 1)                      # the energy steps of mdx_step count the pairs inside the cutoff
        e.step(nsteps)
        x, _, box = e.get_state()
        e.pop_energy_frames()
        e.step(1)                               # evaluated at x on the pruned list
        fr = e.pop_energy_frames()
        st = e.list_stats()
    finally:
        e.close()
    xs = x.astype(np.float64)
    o = R.nonbonded(xs, s.charge, s.sigma, s.eps, np.asarray(box, np.float64), 1.0, e.beta, s.excl_start, s.excl_list, flags=3, nthreads=8)
    assert len(fr) == 1 and int(fr[0, 0]) == nsteps
    if exact:
        assert abs(st["pairs_within_rc"] - o["npairs"]) <= 2, (st["pairs_within_rc"], o["npairs"])
    else:
        # radii from the drift estimate: pairs may slip past either buffer (that is what the tolerance prices), never more
        # than a few in ten thousand, and never one that is not inside the cutoff
        assert 0 <= o["npairs"] - st["pairs_within_rc"] <= 5e-4 * o["npairs"], (st["pairs_within_rc"], o["npairs"])
    assert abs(fr[0, 2 + eng.E["LJ"]] - o["e_lj"]) < 1e-5 * max(abs(o["e_lj"]), 1.0)
    assert abs(fr[0, 2 + eng.E["COUL_SR"]] - o["e_coul_sr"]) < 1e-5 * max(abs(o["e_coul_sr"]), abs(o["e_lj"]), 1.0)


@pytest.mark.parametrize("nsteps", [1, 4, 9, 13])
def test_pruned_list_holds_every_pair_inside_the_cutoff(small_complex, nsteps):
    """Outer list for 12 steps, pruned every 4, with buffers no atom crosses in that time (0.25 nm in 24 fs, 0.1 nm in
    10 fs), so that the pruning itself is what is tested: step 1 runs on the first pruned list, step 4 on its last use,
    step 9 on the list pruned at step 8 (written beside the step that read the one before), step 13 behind an outer
    rebuild. The kernel's count of pairs inside the cutoff equals the oracle's at the same positions."""
    _pruned_step_against_oracle(small_complex, 12, 4, nsteps, rlist=1.25, rlist_inner=1.1)


@pytest.mark.parametrize("nsteps", [4, 23])
def test_pruned_list_with_the_estimated_radii(small_complex, nsteps):
    """The radii the engine derives from `verlet-buffer-tolerance` (outer list 20 steps, pruned every 4)."""
    _pruned_step_against_oracle(small_complex, 20, 4, nsteps, exact=False)


def test_pruned_list_sparse_and_smallest_box():
    s = synth.water_box(box=np.diag([4.0, 4.0, 4.0]), density=0.5)
    s.x = s.x.astype(np.float32).astype(np.float64)
    _pruned_step_against_oracle(s, 12, 3, 7, rlist=1.3, rlist_inner=1.1)
    s = synth.water_box(box=np.diag([2.25, 2.25, 2.25]))
    s.x = s.x.astype(np.float32).astype(np.float64)
    _pruned_step_against_oracle(s, 6, 3, 4, rlist=1.12, rlist_inner=1.08)


def test_dual_list_run_follows_single_list_run(small_complex):
    s = small_complex
    out = {}
    for name, kw in (("single", dict(nstlist=10)), ("dual", dict(nstlist=40, nstprune=5))):
        e = eng.Engine(s, rc=1.0, dt=0.002, nstcalcenergy=0, nstcomm=0, **kw)
        e.step(45)
        out[name] = e.get_state()[0].astype(np.float64)
        e.close()
    d = out["dual"] - out["single"]
    # the same pairs inside the cutoff on both lists: the runs differ by the order of the float additions only
    assert np.sqrt((d * d).sum(1).mean()) < 1e-4


def test_prune_arguments(small_water):
    s = small_water
    e = eng.Engine(s, rc=1.0, nstlist=20, rlist=1.1)
    try:
        with pytest.raises(eng.MdxError, match="rlist_inner"):
            e._call("mdx_set_pairlist_prune", __import__("ctypes").c_float(1.2), 4)
        e._call("mdx_set_pairlist_prune", __import__("ctypes").c_float(1.05), 4)
        e.step(12); e.synchronize()
        e._call("mdx_set_pairlist_prune", __import__("ctypes").c_float(0.0), 0)          # off again
        e.step(25); e.synchronize()
        assert np.isfinite(e.get_state()[0]).all()
    finally:
        e.close()
