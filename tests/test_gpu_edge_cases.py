"""Edge cases through the C ABI: a sparse system (mostly empty search columns, clusters that are mostly filler atoms), the
smallest box the cutoff admits, a box below that, and argument errors."""
import ctypes

import numpy as np
import pytest

import mdx  # noqa: F401
from mdx import synth, engine as eng
from oracle import mdref as R

pytestmark = pytest.mark.gpu


def relrms(a, b):
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


def _against_oracle(s, rlist=None):
    s.x = s.x.astype(np.float32).astype(np.float64)
    e = eng.Engine(s, rc=1.0, rlist=rlist, nstcalcenergy=1)
    try:
        f, en = e.compute(eng.F_NONBONDED | eng.F_EXCL)
        o = R.nonbonded(s.x, s.charge, s.sigma, s.eps, s.box, 1.0, e.beta, s.excl_start, s.excl_list, flags=3, nthreads=4)
        assert relrms(f, o["f"]) < 1e-5
        assert abs(en["LJ"] - o["e_lj"]) < 1e-5 * max(abs(o["e_lj"]), 1.0)
        assert abs(en["COUL_SR"] - o["e_coul_sr"]) < 1e-5 * max(abs(o["e_coul_sr"]), abs(o["e_lj"]), 1.0)
        assert e.list_stats()["pairs_within_rc"] == o["npairs"]
        f, en = e.compute(eng.F_PME)
        Ep, Fp, _ = R.pme_recip(s.x, s.charge, s.box, e.beta, e.grid)
        assert relrms(f, Fp) < 1e-5 and abs(en["COUL_RECIP"] - Ep) < 1e-5 * max(abs(Ep), 1.0)
        e.step(25); e.synchronize()                 # across a list rebuild
        x, _, _ = e.get_state()
        assert np.isfinite(x).all()
        w = s.settles
        d = np.linalg.norm(x[w[:, 0]].astype(np.float64) - x[w[:, 1]], axis=1)
        assert np.abs(d / s.settle_doh - 1).max() < 3e-5
    finally:
        e.close()


def test_sparse_system_with_empty_columns():
    s = synth.water_box(box=np.diag([4.0, 4.0, 4.0]), density=0.5)
    assert s.natoms < 100
    _against_oracle(s)


def test_smallest_box_for_the_cutoff():
    """A cubic box barely above twice the pair-list radius: every atom interacts with images across all faces."""
    s = synth.water_box(box=np.diag([2.25, 2.25, 2.25]))
    _against_oracle(s, rlist=1.1)


def test_box_below_twice_the_list_radius_is_refused():
    s = synth.water_box(box=np.diag([1.9, 1.9, 1.9]))
    with pytest.raises(eng.MdxError, match="shorter than twice"):
        eng.Engine(s, rc=1.0)


def test_argument_errors_are_reported_not_executed():
    s = synth.water_box(box=np.diag([3.0, 3.0, 3.0]))
    e = eng.Engine(s, rc=1.0)
    try:
        a = np.array([[0, s.natoms + 5]], np.int32); b = np.array([0.1], np.float32)
        ip, fp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)
        with pytest.raises(eng.MdxError, match="out of range"):
            e._call("mdx_set_bonds", 1, a.ctypes.data_as(ip), b.ctypes.data_as(fp), b.ctypes.data_as(fp))
        with pytest.raises(eng.MdxError, match="nsteps"):
            e.step(-1)
        e.step(0)
        assert e.get_step() == 0
        # the handle is still usable after a refused call
        e.step(3); e.synchronize()
        assert e.get_step() == 3
    finally:
        e.close()


def test_search_after_the_state_was_replaced():
    """The pair search walks the atoms in the cluster order of the previous search; that order says nothing about a state
    set afterwards (here: every molecule moved to another molecule's place) and must not matter for the result."""
    s = synth.water_box(box=np.diag([3.0, 3.0, 3.0]))
    s.x = s.x.astype(np.float32).astype(np.float64)
    e = eng.Engine(s, rc=1.0, nstcalcenergy=1)
    try:
        e.compute(eng.F_NONBONDED)
        nw = s.natoms // 3
        p = np.random.default_rng(5).permutation(nw)
        x2 = s.x.reshape(nw, 3, 3)[p].reshape(-1, 3)
        e.set_state(x2, s.v, s.box)
        f, en = e.compute(eng.F_NONBONDED | eng.F_EXCL)
        o = R.nonbonded(x2, s.charge, s.sigma, s.eps, s.box, 1.0, e.beta, s.excl_start, s.excl_list, flags=3, nthreads=4)
        assert relrms(f, o["f"]) < 1e-5
        assert e.list_stats()["pairs_within_rc"] == o["npairs"]
    finally:
        e.close()
