"""The engine's own 3D transform (two direct DFT passes per line, influence function fused into the z pass) on grid
shapes with every factor the .mdp grids can have (2, 3, 5, 7; equal and unequal splits; odd sizes), against the
float64 oracle's PME on the same grid (/root/reference/mdp/production.mdp:27-29 switches the mesh part on) and against
the rocFFT plan of the same handle. The per-plane kernels, off by default, are checked the same way."""
import numpy as np
import pytest

import mdx  # noqa: F401
from mdx import engine as eng
from oracle import mdref as R

pytestmark = pytest.mark.gpu


def relrms(a, b):
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


# ((30, 90, 72): a k_x plane of 54 KB, above the default limit of dynamic LDS per workgroup; (22, 26, 24): 26 = 2 x 13 has a
# factor above 10, which the plane kernel takes in its second form (radices up to 16); (22, 34, 24): 34 = 2 x 17 has none
# the plane kernel takes, so the y and z passes run one launch each; (20, 160, 112): the planes of the million-atom box's
# grid, 143 KB with factors 10 x 16 and 8 x 14, in the plane kernel's second form with the LDS limit raised for it)
@pytest.mark.parametrize("grid", [(24, 24, 16), (28, 20, 18), (21, 25, 27), (32, 36, 40), (30, 42, 14), (49, 24, 35), (22, 26, 24), (22, 34, 24),
                                  (30, 90, 72), (20, 160, 112)])
def test_own_transform_matches_oracle_and_library(grid, small_water, monkeypatch):
    s = small_water
    if grid == (20, 160, 112):
        monkeypatch.setenv("MDX_YZ_LDS_KB", "150")
    e = eng.Engine(s, rc=1.0, nstcalcenergy=1, grid=grid)
    try:
        f_own, en_own = e.compute(eng.F_PME)
        v_own, _ = e.virial()
        Ep, Fp, vp = R.pme_recip(s.x, s.charge, s.box, e.beta, grid)
        assert relrms(f_own, Fp) < 1e-5
        assert abs(en_own["COUL_RECIP"] - Ep) < 1e-5 * abs(Ep)
        assert np.abs(v_own - vp).max() < 1e-4 * np.abs(vp).max()
        e.debug_set(2, 512)                 # library plan + stand-alone solve kernel
        f_lib, en_lib = e.compute(eng.F_PME)
        assert relrms(f_own, f_lib) < 3e-6
        assert abs(en_own["COUL_RECIP"] - en_lib["COUL_RECIP"]) < 1e-6 * abs(Ep)
        e.debug_set(2, 0x2000000)           # y and z passes as separate launches (the default fuses them per k_x plane)
        f_pl, en_pl = e.compute(eng.F_PME)
        assert relrms(f_pl, Fp) < 1e-5
        assert abs(en_pl["COUL_RECIP"] - Ep) < 1e-5 * abs(Ep)
        # twice in a row: the x pass must leave the charge grid cleared for the next spreading
        e.debug_set(2, 0)
        f_again, _ = e.compute(eng.F_PME)
        assert relrms(f_again, f_own) < 1e-6
    finally:
        e.close()
