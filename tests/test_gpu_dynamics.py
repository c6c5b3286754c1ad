"""GPU dynamics: update, constraints, coupling and pulling against the float64 oracle and
against conservation / known-answer properties."""
import numpy as np
import pytest

import mdx  # noqa: F401
from mdx import synth, engine as eng
from oracle import mdref as R

pytestmark = pytest.mark.gpu


def _copy_with(s, x, v):
    import copy
    t = copy.copy(s)
    t.x = np.asarray(x, dtype=np.float64); t.v = np.asarray(v, dtype=np.float64)
    return t


def test_one_step_matches_oracle_water(small_water):
    s = small_water
    v0 = s.v.astype(np.float32).astype(np.float64)
    e = eng.Engine(s, rc=1.0, nstcalcenergy=0, nstcomm=0, dt=0.002)
    e.set_state(s.x, v0, s.box)
    e.step(1)
    x1, v1, _ = e.get_state()
    xr, vr, _ = R.md_step(_copy_with(s, s.x, v0), v0, 1.0, e.beta, e.grid, 0.002, nthreads=8)
    e.close()
    assert np.abs(x1 - xr).max() < 3e-6
    assert np.abs(v1 - vr).max() < 2e-3
    d = np.linalg.norm(x1[0::3].astype(np.float64) - x1[1::3], axis=1)
    assert np.abs(d / 0.09572 - 1).max() < 2e-5          # float32 coordinates at ~3 nm


def test_one_step_matches_oracle_complex_lincs(small_complex):
    s = small_complex
    v0 = s.v.astype(np.float32).astype(np.float64)
    e = eng.Engine(s, rc=1.0, nstcalcenergy=0, nstcomm=0, dt=0.002, lincs_order=4, lincs_iter=1)
    e.set_state(s.x, v0, s.box)
    e.step(1)
    x1, v1, _ = e.get_state()
    xr, vr, _ = R.md_step(_copy_with(s, s.x, v0), v0, 1.0, e.beta, e.grid, 0.002, nthreads=8)
    e.close()
    assert np.abs(x1 - xr).max() < 4e-6
    c = s.constraints
    d = np.linalg.norm(x1[c[:, 0]].astype(np.float64) - x1[c[:, 1]], axis=1)
    dr = np.linalg.norm(xr[c[:, 0]] - xr[c[:, 1]], axis=1)
    # same LINCS truncation error as the oracle (order 4, 1 iteration), up to float32 coordinates
    assert np.abs(d - dr).max() < 5e-6
    assert np.abs(d / s.constraint_len - 1).max() < 1e-4


def test_lincs_order_iter_improves_accuracy(small_complex):
    s = small_complex
    devs = []
    for order, it in ((4, 1), (6, 2)):
        e = eng.Engine(s, rc=1.0, nstcalcenergy=0, dt=0.002, lincs_order=order, lincs_iter=it)
        e.step(5)
        x, _, _ = e.get_state(); e.close()
        c = s.constraints
        d = np.linalg.norm(x[c[:, 0]].astype(np.float64) - x[c[:, 1]], axis=1)
        devs.append(np.sqrt(((d / s.constraint_len - 1) ** 2).mean()))
    assert devs[1] <= devs[0] * 1.05 and devs[0] < 5e-5


@pytest.fixture(scope="module")
def equilibrated_water():
    s = synth.water_box(d=4.0)
    e = eng.Engine(s, rc=1.0, dt=0.001, nstcalcenergy=100, tc_groups=["System"], ref_t=[300.0], tau_t=[0.05], nsttcouple=5)
    e.step(6000)
    x, v, box = e.get_state()
    e.close()
    import copy
    t = copy.copy(s); t.x = x.astype(np.float64); t.v = v.astype(np.float64)
    return t


def test_nve_energy_drift(equilibrated_water):
    s = equilibrated_water
    drifts = {}
    for dt in (0.001, 0.002):
        e = eng.Engine(s, rc=1.0, dt=dt, nstlist=10, nstcalcenergy=20, nstcomm=0)
        nsteps = int(round(4.0 / dt))
        e.step(nsteps + 1)
        fr = e.pop_energy_frames(); e.close()
        t = fr[:, 1]; tot = fr[:, 2 + eng.E["POT"]] + fr[:, 2 + eng.E["KIN"]]
        slope = np.polyfit(t, tot, 1)[0] / s.natoms            # kJ/mol/ps/atom
        drifts[dt] = slope
        assert np.isfinite(tot).all()
    # GROMACS' verlet-buffer-tolerance default targets 0.005 kJ/mol/ps/atom
    assert abs(drifts[0.001]) < 0.005 and abs(drifts[0.002]) < 0.02, drifts


def test_vrescale_temperature_and_geometry(equilibrated_water):
    s = equilibrated_water
    e = eng.Engine(s, rc=1.0, dt=0.002, nstcalcenergy=10, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1], nsttcouple=10, seed=7)
    e.step(3001)
    fr = e.pop_energy_frames()
    x, _, _ = e.get_state(); e.close()
    T = fr[50:, 2 + eng.E["TEMP"]]
    ndf = 2 * s.natoms - 3
    # canonical fluctuations: sigma_T / T = sqrt(2/ndf); the mean must sit at 300 K within a few sigma of the mean
    assert abs(T.mean() - 300.0) < 3.0
    assert 0.3 * 300 * np.sqrt(2 / ndf) < T.std() < 3 * 300 * np.sqrt(2 / ndf)
    d1 = np.linalg.norm(x[0::3].astype(np.float64) - x[1::3], axis=1); d3 = np.linalg.norm(x[1::3].astype(np.float64) - x[2::3], axis=1)
    assert np.abs(d1 / 0.09572 - 1).max() < 3e-5 and np.abs(d3 / 0.15139 - 1).max() < 3e-5


@pytest.mark.parametrize("kind", [1, 2])
def test_pressure_coupling_relaxes_density(equilibrated_water, kind):
    s = equilibrated_water
    e = eng.Engine(s, rc=1.0, dt=0.002, nstcalcenergy=10, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1],
                   pcoupl=kind, tau_p=1.0, ref_p=1.0, compressibility=4.5e-5, nstpcouple=10, dispcorr=2)
    e.step(4001)
    fr = e.pop_energy_frames(); e.close()
    vol = fr[:, 2 + eng.E["VOLUME"]]; P = fr[:, 2 + eng.E["PRES"]]
    assert np.isfinite(vol).all() and np.isfinite(P).all()
    dens = (s.natoms / 3) * 18.015 / 6.02214e23 / (vol * 1e-21)   # g/cm^3
    # TIP3P at 300 K / 1 bar: ~0.98 g/cm^3; the lattice start is 33.8 molecules/nm^3 = 1.01 g/cm^3
    assert 0.93 < dens[-20:].mean() < 1.05
    assert vol[-1] != vol[0]
    # pressure fluctuates around the reference: |<P>| far below its instantaneous spread
    assert abs(P[100:].mean() - 1.0) < 0.5 * P[100:].std() + 150


@pytest.mark.parametrize("mode", ["no", "all", "com"])
def test_refcoord_scaling_modes(small_complex, mode):
    """`refcoord_scaling` (/root/reference/mdp/npt_complex.mdp:72 `all`, /root/reference/mdp/npt.mdp:40 `com`): the
    restraint virial of the starting configuration against the oracle, then 40 steps of C-rescale with an exaggerated
    compressibility; the references must have followed the box as the oracle's rule says."""
    s = small_complex
    code = {"no": 0, "all": 1, "com": 2}[mode]
    e = eng.Engine(s, rc=1.0, dt=0.001, nstcalcenergy=10, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1], use_posres=True,
                   refcoord_scaling=code, pcoupl=1, tau_p=0.5, ref_p=1.0, compressibility=2e-3, nstpcouple=10)
    ref0 = s.posres_ref.astype(np.float32).astype(np.float64)
    rng = np.random.default_rng(11)
    x = (s.x + rng.normal(scale=0.004, size=s.x.shape)).astype(np.float32).astype(np.float64)
    e.set_state(x, s.v, s.box)
    e.compute(eng.F_POSRES)
    v, _ = e.virial()
    vo = R.posres(x, s.box, s.posres_idx, ref0, s.posres_k, refcoord_scaling=mode)[2]
    assert np.abs(v - vo).max() < 1e-4 * np.abs(vo).max()
    assert np.abs(e.posres_ref() - ref0).max() == 0.0
    e.set_state(s.x, s.v, s.box)
    e.step(41)
    _, _, box = e.get_state()
    got = e.posres_ref().astype(np.float64); e.close()
    mu = float(box[0, 0]) / float(np.float32(s.box[0, 0]))
    assert abs(mu - 1.0) > 1e-4            # the box did move
    want = R.scale_posres_ref(ref0, mu, mode)
    assert np.abs(got - want).max() < 2e-5
    if mode != "no":
        assert np.abs(got - ref0).max() > 1e-4


def test_constant_force_pull(small_complex):
    """`pull_coord1_type = constant-force`, k = 500, start = yes
    (/root/reference/mdp/pull_short.mdp:41-46): the force column is -k and the COM distance
    matches the oracle at t = 0, then decreases (attraction)."""
    s = small_complex
    e = eng.Engine(s, rc=1.0, dt=0.001, nstcalcenergy=100, tc_groups=["Protein", "Non-Protein"], ref_t=[300, 300], tau_t=[0.1, 0.1])
    e.set_pull(s.groups["ActiveSite"], s.groups["MOL"], type="constant-force", k=5000.0, nstout=50)
    e.step(1501)
    rec = e.pop_pull_records(); e.close()
    r0, _, _, _ = R.pull_distance(s.x, s.box, s.mass, s.groups["ActiveSite"], s.groups["MOL"])
    assert len(rec) == 31
    assert np.allclose(rec[:, 0], np.arange(31) * 0.05)
    assert abs(rec[0, 1] - r0) < 2e-6
    assert np.allclose(rec[:, 2], -5000.0)
    assert rec[-1, 1] < rec[0, 1] - 0.01
    # work of a constant force: W = -k (r - r0)
    assert np.allclose(rec[:, 3], -5000.0 * (rec[:, 1] - rec[0, 1]), atol=1e-6 * 5000)


def test_umbrella_pull_rate(small_complex):
    """Umbrella with a rate (the constant-velocity variant of /root/reference/config.yaml:38-39)."""
    s = small_complex
    e = eng.Engine(s, rc=1.0, dt=0.001, nstcalcenergy=100, tc_groups=["System"], ref_t=[300], tau_t=[0.1])
    e.set_pull(s.groups["ActiveSite"], s.groups["MOL"], type="umbrella", k=1000.0, rate=-0.1, nstout=10)
    e.step(501)
    rec = e.pop_pull_records(); e.close()
    t, r, f = rec[:, 0], rec[:, 1], rec[:, 2]
    ref = r[0] - 0.1 * t
    assert np.allclose(f, -1000.0 * (r - ref), atol=1e-3)
    assert abs(f[0]) < 1e-3
