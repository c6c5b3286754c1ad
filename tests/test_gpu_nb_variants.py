"""The selectable variants of the nonbonded kernel agree with the default: tabulated against analytic Ewald correction
(which also sums the virial pair by pair where the tabulated kernel uses position-force sums), and ticket queues against
the static deal of the work units.
(The default itself is checked against the oracle in test_gpu_parity.py.)"""
import numpy as np
import pytest

import mdx  # noqa: F401
from mdx import engine as eng

pytestmark = pytest.mark.gpu


def relrms(a, b):
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


def test_variants_agree(small_complex):
    s = small_complex
    e = eng.Engine(s, rc=1.0, nstcalcenergy=1)
    try:
        f0, en0 = e.compute(eng.F_NONBONDED | eng.F_EXCL)
        v0, _ = e.virial()
        for key, val, label in ((2, 0x400000, "analytic"), (3, 0, "static deal"), (3, 3, "three dynamic rounds")):
            e.debug_set(key, val)
            f, en = e.compute(eng.F_NONBONDED | eng.F_EXCL)
            v, _ = e.virial()
            e.debug_set(key, 0 if key == 2 else 1)
            assert relrms(f, f0) < 3e-6, label
            assert np.abs(v - v0).max() < 2e-5 * np.abs(v0).max(), label
            # (the short-range Coulomb sum nearly cancels in a neutral system: the scale is the largest of the terms)
            scale = max(abs(en0[t]) for t in ("LJ", "COUL_SR", "COUL_EXCL"))
            for term in ("LJ", "COUL_SR", "COUL_EXCL"):
                assert abs(en[term] - en0[term]) < 1e-5 * scale, (label, term, en[term], en0[term])
        # force-only launches (the step loop) against the energy launch: 40 steps from identical states
        x0, v0, box = e.get_state()
        e.step(40); xa, _, _ = e.get_state()
        e.set_state(x0, v0, box); e.set_step(0); e.debug_set(2, 0x400000)
        e.step(40); xb, _, _ = e.get_state()
        assert np.abs(xa - xb).max() < 5e-4
    finally:
        e.close()


def test_force_kernel_with_type_table_agrees_with_per_pair_combination(small_complex):
    """The force-only kernel of the step loop takes (12 C12, 6 C6) of a pair from a table by atom type; mdx_debug_set key 2
    bit 0x1000000 selects the kernel that combines half-sigmas and well depths per pair (also what a system with more
    types than the table holds, and a decomposed run, get). One step from rest, so the new velocities are the forces."""
    import copy
    s = copy.copy(small_complex)
    s.v = np.zeros_like(s.x)
    out = []
    for bit in (0, 0x1000000):
        e = eng.Engine(s, rc=1.0, nstcalcenergy=0, dt=0.002)
        e.debug_set(2, bit)
        e.step(1)
        _, v, _ = e.get_state()
        out.append(v.astype(np.float64))
        e.close()
    assert relrms(out[0], out[1]) < 1e-5
    assert np.abs(out[0]).max() > 1e-3


def test_force_only_kernel_agrees_with_the_energy_kernel(small_complex):
    """The step loop's force-only kernel (plain and general list entries walked in two loops of their own, Lennard-Jones by
    type table) against the energy kernel, which test_gpu_parity.py checks against the oracle: one step from rest with and
    without energies on that step, so the new velocities are the forces of the two kernels behind identical updates."""
    import copy
    s = copy.copy(small_complex)
    s.v = np.zeros_like(s.x)
    out = []
    for nstcalc in (0, 1):
        e = eng.Engine(s, rc=1.0, nstcalcenergy=nstcalc, dt=0.002)
        e.step(1)
        _, v, _ = e.get_state()
        out.append(v.astype(np.float64))
        e.close()
    assert relrms(out[0], out[1]) < 1e-5
    assert np.abs(out[0]).max() > 1e-3
