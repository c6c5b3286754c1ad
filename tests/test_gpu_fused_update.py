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
