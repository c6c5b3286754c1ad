"""Start-up and run-control entry points of the C ABI on the GPU: velocity generation
(`gen_vel`, /root/reference/mdp/nvt_complex.mdp:39-41), constraining the start (`continuation = no`,
nvt_complex.mdp:13), steepest descent (/root/reference/mdp/em_complex.mdp:3-6), the step counter
used by checkpoint restarts, and the asynchronous trajectory-frame ring."""
import os

import numpy as np
import pytest

import mdx  # noqa: F401
from mdx import engine as eng, units as U
from oracle import mdref as R

pytestmark = pytest.mark.gpu


def test_generated_velocities_are_maxwell_boltzmann(small_complex):
    s = small_complex
    e = eng.Engine(s, rc=1.0, nstcalcenergy=1)
    e.generate_velocities(300.0, 12345)
    _, v, _ = e.get_state()
    v = v.astype(np.float64)
    m = s.mass
    # no centre-of-mass motion, exactly 300 K over 3N degrees of freedom
    assert np.abs((m[:, None] * v).sum(0) / m.sum()).max() < 1e-4
    ke = 0.5 * (m[:, None] * v * v).sum()
    assert abs(2 * ke / (3 * s.natoms * U.BOLTZ) - 300.0) < 0.05
    # component distribution: unit variance in units of sqrt(kT/m), light and heavy atoms alike
    z = v / np.sqrt(U.BOLTZ * 300.0 / m)[:, None]
    assert abs(z.std() - 1.0) < 0.02 and abs(z.mean()) < 0.02
    assert abs((z ** 4).mean() - 3.0) < 0.15
    light = m < 2.0
    assert abs(z[light].std() - z[~light].std()) < 0.03
    # reproducible per seed, different across seeds
    e.generate_velocities(300.0, 12345)
    _, v2, _ = e.get_state()
    e.generate_velocities(300.0, 12346)
    _, v3, _ = e.get_state()
    e.close()
    assert np.array_equal(v2.astype(np.float64), v) and not np.allclose(v3, v2)


def test_constrain_start_projects_positions_and_velocities(small_complex):
    s = small_complex
    rng = np.random.default_rng(5)
    x = (s.x + rng.normal(scale=0.003, size=s.x.shape)).astype(np.float32).astype(np.float64)
    e = eng.Engine(s, rc=1.0, nstcalcenergy=1, lincs_iter=2, lincs_order=6)
    e.set_state(x, s.v, s.box)
    e.generate_velocities(300.0, 7)
    e.constrain_start()
    xc, vc, _ = e.get_state()
    e_dt = e.dt
    e.close()
    xc = xc.astype(np.float64); vc = vc.astype(np.float64)
    c = s.constraints
    d = xc[c[:, 0]] - xc[c[:, 1]]
    assert np.abs(np.linalg.norm(d, axis=1) / s.constraint_len - 1.0).max() < 2e-5
    # leap-frog start: the velocities are those that carry the constrained x(-dt) to x(0), so the
    # back-stepped positions satisfy the constraints too (this leaves the centripetal component
    # along a bond, v_perp^2 dt / 2L, in place)
    xb = xc - e_dt * vc
    db = xb[c[:, 0]] - xb[c[:, 1]]
    assert np.abs(np.linalg.norm(db, axis=1) / s.constraint_len - 1.0).max() < 5e-5
    w = s.settles
    for a, b, L in ((0, 1, s.settle_doh), (0, 2, s.settle_doh), (1, 2, s.settle_dhh)):
        dw = xc[w[:, a]] - xc[w[:, b]]
        assert np.abs(np.linalg.norm(dw, axis=1) / L - 1.0).max() < 2e-5
        dwb = xb[w[:, a]] - xb[w[:, b]]
        assert np.abs(np.linalg.norm(dwb, axis=1) / L - 1.0).max() < 5e-5
        rel = ((vc[w[:, a]] - vc[w[:, b]]) * dw).sum(1) / L
        vrel2 = ((vc[w[:, a]] - vc[w[:, b]]) ** 2).sum(1)
        assert np.abs(rel - 0.5 * e_dt * vrel2 / L).max() < 0.02
    # the projection moved atoms by no more than the distortion put in
    assert np.abs(xc - x).max() < 0.02


def test_steepest_descent_lowers_energy_monotonically(small_complex):
    s = small_complex
    rng = np.random.default_rng(9)
    x = (s.x + rng.normal(scale=0.01, size=s.x.shape)).astype(np.float32).astype(np.float64)
    e = eng.Engine(s, rc=1.0, nstcalcenergy=1, use_posres=True)
    e.set_state(x, None, s.box)
    trace = []
    res = e.minimize(40, emtol=10.0, emstep=0.01, progress=lambda st, ep, fm, ss: trace.append((st, ep, fm, ss)))
    xm, _, _ = e.get_state()
    _, en = e.compute()
    e.close()
    ep = np.array([t[1] for t in trace])
    assert res["steps"] == 40 and res["status"] == 0
    assert np.all(np.diff(ep) <= 1e-6 * np.abs(ep[:-1]))           # accepted energies never rise
    assert res["epot"] < res["epot0"] and res["fmax"] < res["fmax0"]
    assert abs(en["POT"] - res["epot"]) < 1e-6 * abs(res["epot"])
    # waters stay rigid during minimisation
    w = s.settles
    xm = xm.astype(np.float64)
    assert np.abs(np.linalg.norm(xm[w[:, 0]] - xm[w[:, 1]], axis=1) - s.settle_doh).max() < 2e-5
    # the minimiser's first energy is the oracle's energy of the same start geometry after constraining
    assert np.isfinite(res["epot0"])


def test_step_counter_and_frame_ring(small_water):
    s = small_water
    e = eng.Engine(s, rc=1.0, nstcalcenergy=10, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1], seed=3)
    e.frame_ring(3)
    e.step(7)
    e.capture_frame(0, want_v=True)
    e.step(5)
    e.capture_frame(1)
    e.step(3)
    st0, x0, v0, b0 = e.frame_wait(0)
    x0 = x0.copy(); v0 = v0.copy()
    st1, x1, v1, b1 = e.frame_wait(1)
    x1 = x1.copy()
    assert (st0, st1) == (7, 12) and v1 is None and np.allclose(b0, s.box, atol=1e-6)
    # a run captured frame-by-frame equals the same run read back synchronously
    e2 = eng.Engine(s, rc=1.0, nstcalcenergy=10, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1], seed=3)
    e2.step(7)
    xa, va, _ = e2.get_state()
    e2.step(5)
    xb, vb, _ = e2.get_state()
    # (force accumulation uses float atomics, so two runs agree to rounding, not bit for bit)
    close = lambda a, b, tol: np.abs(a.astype(np.float64) - b).max() < tol
    assert close(xa, x0, 1e-5) and close(va, v0, 5e-3) and close(xb, x1, 1e-5)
    e2.step(3)
    xc, vc, bc = e2.get_state()
    xe, ve, _ = e.get_state()
    assert e.get_step() == 15 and close(xe, xc, 2e-5)
    # restart: state + step counter continue the trajectory (a fresh pair list changes only the
    # summation order of the forces)
    e3 = eng.Engine(s, rc=1.0, nstcalcenergy=10, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1], seed=3)
    e3.set_state(xb, vb, s.box)
    e3.set_step(12)
    e3.step(3)
    xr, vr, _ = e3.get_state()
    assert e3.get_step() == 15
    assert np.abs(xr - xc).max() < 2e-5 and np.abs(vr - vc).max() < 5e-3
    e.close(); e2.close(); e3.close()
    with pytest.raises(eng.MdxError):
        e4 = eng.Engine(s, rc=1.0)
        try:
            e4.capture_frame(0)
        finally:
            e4.close()


def test_xtc_capture_rounds_on_the_gpu_like_the_host_encoder(small_complex, tmp_path):
    """`nstxout-compressed` frames (/root/reference/mdp/production.mdp:12-13) captured as xtc integers: the same state
    captured both ways gives the same integers as the host rounding, the same frame statistics, and a byte-identical
    frame; then the output pipeline with device-rounded frames against the same run written from float captures."""
    from mdx.io import traj
    from mdx.output import FramePipeline
    from test_io_formats import xtc_round_like_the_capture_kernel
    s = small_complex
    e = eng.Engine(s, rc=1.0, nstcalcenergy=0, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1], seed=3)
    e.frame_ring(4)
    e.step(9)
    e.capture_frame(0)
    e.capture_frame_xtc(1, 1000.0, want_v=True)
    e.step(2)
    st0, x, _, box0 = e.frame_wait(0)
    st1, ix, v, box1, stats = e.frame_wait_xtc(1)
    assert st0 == st1 == 9 and v is not None and np.array_equal(box0, box1)
    want_ix, want_stats = xtc_round_like_the_capture_kernel(x)
    assert np.array_equal(ix, want_ix)
    assert np.array_equal(stats, want_stats)
    w = traj.XtcWriter(os.devnull, s.natoms, threads=2)
    assert w.encode_quantized(st1, 0.018, box1, ix, stats) == w.encode(st0, 0.018, box0, x)
    with pytest.raises(eng.MdxError):
        e.frame_wait(1)                  # a slot of integers is not handed out as floats
    e.close()
    # the pipeline: frames every 5 steps, device-rounded (capture) against float captures (submit), same run
    files = []
    for quant in (True, False):
        e = eng.Engine(s, rc=1.0, nstcalcenergy=0, nstlist=10)
        e.frame_ring(4)
        p = str(tmp_path / ("q%d.xtc" % quant))
        xw = traj.XtcWriter(p, s.natoms, threads=2)
        pipe = FramePipeline(e, 4, xw, None, None, 0.002, 0.0, nthreads=2)
        for _ in range(4):
            if quant:
                pipe.capture(True, False, False)
            else:
                slot = pipe.acquire_slot(); e.capture_frame(slot); pipe.submit(slot, True, False, False)
            e.step(5)
        e.synchronize(); pipe.finish(); xw.close(); e.close()
        assert pipe.error is None
        files.append(traj.read_xtc(p))
    assert [f["step"] for f in files[0]] == [0, 5, 10, 15] == [f["step"] for f in files[1]]
    assert np.array_equal(files[0][0]["x"], files[1][0]["x"])                 # the same start state, rounded on either side
    for a, b in zip(files[0], files[1]):
        assert np.abs(a["x"] - b["x"]).max() < 2.1e-3                         # two runs: float atomics, then rounding to 1e-3


def test_deferred_capture_edges(small_water):
    """A capture is launched by the next step, beside its force kernels. The cases around that: a wait that no step
    serves, a state replaced between the capture and the wait, a capture on a step that changes the cell (the frame
    carries the cell the positions belong to), and a slot captured into twice with nothing in between."""
    s = small_water
    kw = dict(rc=1.0, nstcalcenergy=10, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1], seed=3)
    e = eng.Engine(s, **kw)
    e.frame_ring(3)
    e.step(4)
    xa, va, ba = e.get_state()
    # (1) capture, then wait without any step: the waiting side launches the capture itself
    e.capture_frame(0, want_v=True)
    st, x, v, b = e.frame_wait(0)
    assert st == 4 and np.array_equal(x, xa) and np.array_equal(v, va)
    # (2) the same slot twice with nothing in between is refused; after a synchronisation it is free again
    e.capture_frame(1)
    with pytest.raises(eng.MdxError):
        e.capture_frame(1)
    e.synchronize()
    e.capture_frame(1)
    # (3) the state replaced behind a capture: the frame still holds the state of the capture
    e.capture_frame_xtc(2, 1000.0)
    e.set_state(xa + 0.5, va, ba)
    _, x1, _, _ = e.frame_wait(1)
    _, ix, _, _, _ = e.frame_wait_xtc(2)
    assert np.array_equal(x1, xa)
    assert np.abs(ix / 1000.0 - xa).max() < 5.1e-4
    e.close()
    # (4) a frame ahead of a step that rescales the cell (C-rescale every 2 steps): cell and positions of the same moment
    e = eng.Engine(s, pcoupl=1, tau_p=0.2, ref_p=1.0, compressibility=4.5e-4, nstpcouple=2, **kw)
    e.frame_ring(2)
    e.step(6)
    xb, _, bb = e.get_state()
    e.capture_frame(0)
    e.step(4)
    _, _, bc = e.get_state()
    st, x, _, b = e.frame_wait(0)
    # (get_state hands the cell out in single precision)
    assert st == 6 and np.array_equal(x, xb) and np.abs(b - bb).max() < 3e-7
    assert np.abs(np.asarray(bc) - np.asarray(bb)).max() > 3e-6          # the cell did move afterwards
    e.close()
