"""GPU coupling algorithms against the float64 oracle fed the same inputs and the same Philox draws:
V-rescale (`tcoupl`, /root/reference/mdp/production.mdp:31-34), C-rescale and Parrinello-Rahman (`pcoupl`,
production.mdp:36-40; npt_complex.mdp:47-52), the pressure tensor they act on, the dispersion correction
(`DispCorr = EnerPres`, production.mdp:44), centre-of-mass motion removal (npt_complex.mdp:64-66), velocity
generation (nvt_complex.mdp:39-41) and the first point of the steepest-descent minimiser (em_complex.mdp:3-6)."""
import copy

import numpy as np
import pytest

import mdx  # noqa: F401
from mdx import engine as eng, units as U
from oracle import mdref as R

pytestmark = pytest.mark.gpu


def _with(s, x=None, v=None):
    t = copy.copy(s)
    if x is not None:
        t.x = np.asarray(x, dtype=np.float64)
    if v is not None:
        t.v = np.asarray(v, dtype=np.float64)
    return t


def _group_ke(s, v, idx):
    return float(0.5 * (s.mass[idx, None] * v[idx] * v[idx]).sum())


def test_vrescale_lambda_and_update_match_oracle(small_complex):
    """The scaling factor the device draws for each T-coupling group equals the oracle's for the same kinetic
    energy and the same (seed, group, step) stream, and it scales the old velocity only: v' = lambda v + f/m dt.
    Reference temperature far from the actual one and a short tau_t make lambda differ from 1 by tens of percent."""
    s = small_complex
    groups, ref_t, tau_t, nst, dt, seed = ["Protein", "Non-Protein"], [600.0, 150.0], [0.01, 0.02], 5, 0.002, 424242
    v0 = s.v.astype(np.float32).astype(np.float64)
    e = eng.Engine(s, rc=1.0, dt=dt, nstcalcenergy=0, nstcomm=0, tc_groups=groups, ref_t=ref_t, tau_t=tau_t, nsttcouple=nst, seed=seed)
    e.set_state(s.x, v0, s.box)
    e.step(1)
    rs = e.get_run_state()
    x1, v1, _ = e.get_state()
    lam_atom = np.ones(s.natoms)
    for g, name in enumerate(groups):
        idx = s.groups[name]
        ek = _group_ke(s, v0, idx)
        assert abs(rs[eng.RS["EKINH"] + g] - ek) < 2e-6 * ek
        lam, eknew = R.vrescale_lambda(ek, ref_t[g], tau_t[g], e.ndf[g], nst * dt, seed, g, 0)
        assert abs(lam - 1.0) > 0.05                       # the test can tell where lambda is applied
        assert abs(rs[eng.RS["LAMBDA"] + g] - lam) < 1e-5 * lam, (g, rs[eng.RS["LAMBDA"] + g], lam)
        # thermostat work: minus the kinetic energy it put in
        assert abs(rs[eng.RS["THERM_INTEGRAL"] + g] + (eknew - ek)) < 1e-5 * abs(eknew - ek)
        lam_atom[idx] = lam
    xr, vr, _ = R.md_step(_with(s, s.x, v0), v0, 1.0, e.beta, e.grid, dt, nthreads=8, lam=lam_atom)
    e.close()
    assert np.abs(x1 - xr).max() < 5e-6
    assert np.abs(v1 - vr).max() < 2.5e-3
    # the same step with lambda applied after the kick differs by (lambda - 1) f/m dt: far outside the tolerance
    f_over_m = (vr - lam_atom[:, None] * v0)
    assert np.abs((lam_atom[:, None] - 1.0) * f_over_m).max() > 0.05


def _oracle_pressure(s, v0, e, dt, dispcorr):
    """Pressure tensor (bar) of one leap-frog step from the oracle: average of the two half-step kinetic tensors
    minus the virial of every force term and of the constraint forces m dx / dt^2, plus the dispersion term."""
    xr, vr, out = R.md_step(_with(s, s.x, v0), v0, 1.0, e.beta, e.grid, dt, nthreads=8)
    m = s.mass
    ekt = 0.25 * (np.einsum("n,na,nb->ab", m, v0, v0) + np.einsum("n,na,nb->ab", m, vr, vr))
    fc = m[:, None] * out["dx_constr"] / dt ** 2
    vir = out["vir"] - 0.5 * np.einsum("na,nb->ab", s.x, fc)
    vol = abs(np.linalg.det(s.box))
    P = 2.0 / vol * (ekt - vir) * R.PRESFAC
    if dispcorr == 2:
        P = P + R.dispersion_correction(s.sigma, s.eps, s.box, 1.0)[1] * np.eye(3)
    return P, ekt.trace(), vir


def test_pressure_tensor_matches_oracle(small_water):
    """What the barostats act on: P = 2/V (E_kin - Xi) with the virial of the nonbonded, mesh and constraint
    forces and the `DispCorr = EnerPres` term."""
    s = small_water
    dt = 0.002
    v0 = s.v.astype(np.float32).astype(np.float64)
    e = eng.Engine(s, rc=1.0, dt=dt, nstcalcenergy=1, nstcomm=0, dispcorr=2)
    e.set_state(s.x, v0, s.box)
    e.step(1)
    vir_d, pres_d = e.virial()
    en = e.energies()
    P, ekin, vir = _oracle_pressure(s, v0, e, dt, 2)
    e.close()
    assert abs(en["KIN"] - ekin) < 1e-5 * ekin
    assert np.abs(vir_d - vir).max() < 2e-4 * np.abs(vir).max()
    # the pressure is a small difference of large terms: tolerance relative to the kinetic part 2 Ekin / 3V
    scale = 2.0 / abs(np.linalg.det(s.box)) * ekin / 3.0 * R.PRESFAC
    assert np.abs(pres_d - P).max() < 3e-4 * scale
    assert abs(en["PRES"] - P.trace() / 3.0) < 3e-4 * scale


def test_crescale_scaling_matches_oracle(small_water):
    """One C-rescale coupling step: the device's mu against Bernetti & Bussi's formula fed the device's own
    pressure, the volume before the step and the Gaussian of the (seed, step) stream; the box follows mu."""
    s = small_water
    dt, seed, tau_p, comp, nstpc = 0.002, 99, 0.5, 4.5e-5, 1
    for step0 in (0, 7):
        e = eng.Engine(s, rc=1.0, dt=dt, nstcalcenergy=1, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1], pcoupl=1, tau_p=tau_p,
                       ref_p=1.0, compressibility=comp, nstpcouple=nstpc, seed=seed, dispcorr=2)
        if step0:
            e.set_step(step0)
        e.step(1)
        _, pres = e.virial()
        rs = e.get_run_state()
        x1, v1, box1 = e.get_state()
        e.close()
        box0 = np.asarray(s.box, dtype=np.float32).astype(np.float64)
        g = R.crescale_gauss(seed, step0)
        mu = R.crescale_mu(np.trace(pres) / 3.0, 1.0, comp, tau_p, nstpc * dt, abs(np.linalg.det(box0)), U.BOLTZ * 300.0, g)
        mu_dev = rs[eng.RS["MU"]]
        assert abs(mu - 1.0) > 1e-7
        assert abs(mu_dev - mu) < 1e-6 * abs(mu - 1.0) + 1e-12, (mu_dev, mu)
        assert np.abs(box1 - mu * box0).max() < 1e-6


def test_crescale_work_enters_the_conserved_quantity(small_water):
    """No thermostat, C-rescale with a short time constant on a box that starts far from the reference pressure: the
    barostat changes the volume by several per cent and the total energy with it; the conserved quantity (total energy
    minus the energy put in by every scaling) must stay put where the total energy does not."""
    s = small_water
    e = eng.Engine(s, rc=1.0, dt=0.001, nstcalcenergy=5, pcoupl=1, tau_p=0.5, ref_p=2000.0, compressibility=4.5e-5, nstpcouple=5, seed=5)
    e.step(401)
    fr = e.pop_energy_frames()
    e.close()
    etot = fr[:, 2 + eng.E["POT"]] + fr[:, 2 + eng.E["KIN"]]
    cons = fr[:, 2 + eng.E["CONSERVED"]]
    vol = fr[:, 2 + eng.E["VOLUME"]]
    assert abs(vol[-1] / vol[0] - 1.0) > 0.005           # the barostat did act
    d_tot, d_cons = etot[-1] - etot[1], cons[-1] - cons[1]
    assert abs(d_tot) > 20.0, d_tot
    assert abs(d_cons) < 0.2 * abs(d_tot), (d_cons, d_tot)


def test_parrinello_rahman_box_follows_oracle(small_complex):
    """The isotropic Parrinello-Rahman box ODE in a non-cubic (dodecahedral) cell: the oracle advances (box, box
    velocity) with the device's pressure tensor of each step; box and box velocity must agree all along."""
    s = small_complex
    dt, tau_p, comp = 0.002, 0.5, 4.5e-5
    e = eng.Engine(s, rc=1.0, dt=dt, nstcalcenergy=1, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1], pcoupl=2, tau_p=tau_p,
                   ref_p=1.0, compressibility=comp, nstpcouple=1, dispcorr=2)
    box = np.asarray(s.box, dtype=np.float32).astype(np.float64)
    boxv = np.zeros((3, 3))
    for k in range(60):
        e.step(1)
        _, pres = e.virial()
        rs = e.get_run_state()
        box, boxv, mu, M = R.parrinello_rahman_step(box, boxv, pres, 1.0, comp, tau_p, dt)
        bv_dev = rs[eng.RS["PR_BOXV"]:eng.RS["PR_BOXV"] + 9].reshape(3, 3)
        assert abs(rs[eng.RS["MU"]] - mu) < 1e-9, k
        assert np.abs(bv_dev - boxv).max() < 1e-6 * np.abs(boxv).max() + 1e-12, k
    _, _, box_dev = e.get_state()
    e.close()
    assert np.abs(boxv).max() > 0 and np.abs(box - np.asarray(s.box)).max() > 1e-7        # the box did move
    assert np.abs(box_dev - box).max() < 1e-6


def test_dispersion_correction_energy_and_pressure(small_complex):
    s = small_complex
    res = {}
    for mode in (0, 1, 2):
        e = eng.Engine(s, rc=1.0, nstcalcenergy=1, dispcorr=mode)
        _, en = e.compute(eng.F_NONBONDED)
        _, pres = e.virial()
        res[mode] = (en["DISPCORR"], pres.copy(), en["POT"])
        e.close()
    e_disp, p_disp = R.dispersion_correction(s.sigma, s.eps, s.box, 1.0)
    assert e_disp < 0 and p_disp < 0
    assert res[0][0] == 0.0
    for mode in (1, 2):
        assert abs(res[mode][0] - e_disp) < 1e-5 * abs(e_disp)
        assert abs((res[mode][2] - res[0][2]) - e_disp) < 1e-5 * abs(e_disp) + 1e-7 * abs(res[0][2])
    # `Ener` leaves the pressure alone, `EnerPres` shifts its diagonal by the analytic tail term
    assert np.abs(res[1][1] - res[0][1]).max() < 1e-6 * np.abs(res[0][1]).max()
    d = res[2][1] - res[0][1]
    assert np.abs(np.diag(d) - p_disp).max() < 1e-5 * abs(p_disp)
    # (two separate evaluations: the off-diagonal elements differ by the float32 summation noise of the pair virial, whose
    # order of accumulation is not fixed; 3e-4 bar has been seen)
    assert np.abs(d - np.diag(np.diag(d))).max() < 5e-6 * abs(p_disp)


def test_com_motion_removal_matches_oracle(small_water):
    """Two runs of one step from the same state, one of them removing the centre-of-mass motion at the end of the
    step (`nstcomm = 1`): its velocities are the other's minus sum(m v) / sum(m)."""
    s = small_water
    v0 = (s.v + np.array([0.05, -0.02, 0.03])).astype(np.float32).astype(np.float64)
    out = {}
    for nstcomm in (0, 1):
        e = eng.Engine(s, rc=1.0, dt=0.002, nstcalcenergy=0, nstcomm=nstcomm)
        e.set_state(s.x, v0, s.box)
        e.step(1)
        _, v, _ = e.get_state()
        out[nstcomm] = v.astype(np.float64)
        if nstcomm:
            rs = e.get_run_state()
        e.close()
    want = R.remove_com_velocity(out[0], s.mass)
    vcom = (s.mass[:, None] * out[0]).sum(0) / s.mass.sum()
    assert np.abs(vcom).max() > 0.01
    # the momentum sum the device formed (px, py, pz, mass)
    com = rs[eng.RS["COM_V"]:eng.RS["COM_V"] + 4]
    assert abs(com[3] - s.mass.sum()) < 1e-6 * s.mass.sum()
    assert np.abs(com[:3] / com[3] - vcom).max() < 1e-5 * np.abs(vcom).max()
    # two runs agree to the float32 resolution of the constrained positions over dt (~2e-7 nm / 0.002 ps), so the
    # subtraction itself is pinned by the mean difference, which that noise does not touch
    assert np.abs(out[1] - want).max() < 2.5e-4
    assert np.abs((out[1] - want).mean(0)).max() < 1e-5 * np.abs(vcom).max()
    assert np.abs((s.mass[:, None] * out[1]).sum(0) / s.mass.sum()).max() < 1e-6


def test_generated_velocities_match_oracle(small_complex):
    s = small_complex
    e = eng.Engine(s, rc=1.0, nstcalcenergy=1)
    for seed in (12345, 2 ** 40 + 17):
        e.generate_velocities(300.0, seed)
        _, v, _ = e.get_state()
        want = R.generate_velocities(s.mass, 300.0, seed)
        err = np.sqrt(((v - want) ** 2).sum() / (want ** 2).sum())
        assert err < 1e-5, err
        assert np.abs(v - want).max() < 1e-5 * np.abs(want).max()
    e.close()


def test_minimizer_first_point_matches_oracle(small_complex):
    """`integrator = steep` starts from the constrained start geometry: its first potential energy and largest
    force are the oracle's at that geometry (all terms, position restraints and the dispersion correction)."""
    s = small_complex
    rng = np.random.default_rng(9)
    x = (s.x + rng.normal(scale=0.004, size=s.x.shape)).astype(np.float32).astype(np.float64)
    e = eng.Engine(s, rc=1.0, nstcalcenergy=1, use_posres=True, dispcorr=1)
    e.set_state(x, None, s.box)
    res = e.minimize(0, emtol=10.0, emstep=0.01)
    xc, _, _ = e.get_state()
    e.close()
    xc = xc.astype(np.float64)
    t = _with(s, xc)
    ref = R.all_forces(t, 1.0, e.beta, e.grid, nthreads=8)
    ref0 = s.posres_ref.astype(np.float32).astype(np.float64)
    epr, fpr, _ = R.posres(xc, s.box, s.posres_idx, ref0, s.posres_k)
    nb = ref["nb"]
    terms = [nb["e_lj"], nb["e_coul_sr"], nb["e_excl"], ref["recip"][0], ref["bonds"][0], ref["angles"][0], ref["dihedrals"][0],
             ref["pairs14"][0], ref["pairs14"][1], epr, R.dispersion_correction(s.sigma, s.eps, s.box, 1.0)[0]]
    epot = float(sum(terms))
    assert abs(res["epot0"] - epot) < 1e-5 * sum(abs(t_) for t_ in terms)
    fmax = float(np.sqrt(((ref["f"] + fpr) ** 2).sum(1)).max())
    assert abs(res["fmax0"] - fmax) < 1e-4 * fmax
    assert res["steps"] == 0 and res["status"] == 0


def test_restart_carries_coupling_and_pull_state(small_complex):
    """`gmx mdrun -cpi` (/root/reference/scripts/run_smd_with_recovery.py:144-156): a run stopped half way and
    continued from (x, v, box, step, run state) follows the uninterrupted run: the umbrella reference keeps moving
    from its original start (no jump by rate * t), the pull work carries on, the Parrinello-Rahman box velocity and
    the thermostat factors are those of the interrupted run."""
    s = small_complex
    kw = dict(rc=1.0, dt=0.002, nstcalcenergy=10, tc_groups=["Protein", "Non-Protein"], ref_t=[300.0, 300.0], tau_t=[0.1, 0.1],
              nsttcouple=10, pcoupl=2, tau_p=1.0, ref_p=1.0, compressibility=4.5e-5, nstpcouple=10, seed=5, dispcorr=2)
    pull = dict(type="umbrella", k=1000.0, rate=-0.5, nstout=5, start=True)
    n_half = 20

    a = eng.Engine(s, **kw)
    a.set_pull(s.groups["ActiveSite"], s.groups["MOL"], **pull)
    a.step(2 * n_half)
    rec_a = a.pop_pull_records(); rs_a = a.get_run_state(); _, _, box_a = a.get_state(); a.close()

    b = eng.Engine(s, **kw)
    b.set_pull(s.groups["ActiveSite"], s.groups["MOL"], **pull)
    b.step(n_half)
    rec_1 = b.pop_pull_records(); rs = b.get_run_state(); x, v, box = b.get_state(); b.close()
    assert rs[eng.RS["PULL_HAVE_INIT"]] == 1.0 and abs(rs[eng.RS["PR_BOXV"]]) > 0

    c = eng.Engine(_with(_box(s, box), x, v), **kw)
    c.set_step(n_half)
    c.set_pull(s.groups["ActiveSite"], s.groups["MOL"], **pull)
    c.set_run_state(rs)
    rs_back = c.get_run_state()
    for key, n in (("LAMBDA", 8), ("THERM_INTEGRAL", 8), ("PR_BOXV", 9), ("PULL_INIT", 1), ("PULL_WORK", 1), ("BARO_INTEGRAL", 1)):
        o = eng.RS[key]
        assert np.array_equal(rs_back[o:o + n], rs[o:o + n]), key
    c.step(n_half)
    rec_2 = c.pop_pull_records(); rs_c = c.get_run_state(); _, _, box_c = c.get_state(); c.close()

    rec_b = np.concatenate([rec_1, rec_2])
    assert rec_b.shape == rec_a.shape and np.allclose(rec_b[:, 0], rec_a[:, 0], atol=1e-9)
    # the two trajectories differ by rounding only (float atomics, float32 box in the restart)
    assert np.abs(rec_b[:, 1] - rec_a[:, 1]).max() < 2e-5
    assert np.abs(rec_b[:, 2] - rec_a[:, 2]).max() < 1000.0 * 2e-5 + 1e-3          # a reference re-seeded at the restart would be off by k rate t = 20
    assert np.abs(rec_b[:, 3] - rec_a[:, 3]).max() < 1e-3 + 1e-3 * np.abs(rec_a[:, 3]).max()
    assert abs(rs_c[eng.RS["PULL_INIT"]] - rs_a[eng.RS["PULL_INIT"]]) < 1e-12
    bv_a, bv_c = rs_a[eng.RS["PR_BOXV"]:eng.RS["PR_BOXV"] + 9], rs_c[eng.RS["PR_BOXV"]:eng.RS["PR_BOXV"] + 9]
    assert np.abs(bv_c - bv_a).max() < 2e-3 * np.abs(bv_a).max()
    assert np.abs(box_c - box_a).max() < 1e-6
    la, lc = rs_a[eng.RS["LAMBDA"]:eng.RS["LAMBDA"] + 2], rs_c[eng.RS["LAMBDA"]:eng.RS["LAMBDA"] + 2]
    assert np.abs(lc - la).max() < 1e-5


def _box(s, box):
    t = copy.copy(s)
    t.box = np.asarray(box, dtype=np.float64)
    return t
