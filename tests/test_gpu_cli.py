"""The `gmx grompp` / `gmx mdrun` shim driven as the pipeline drives GROMACS: one child process per
tool, stage outputs feeding the next stage, the SMD stage tailed and terminated by SIGTERM
(/root/reference/Snakefile:673-743, /root/reference/scripts/run_smd_with_distance_monitoring.py:86-179)."""
import os
import signal
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GMX = os.path.join(ROOT, "bin", "gmx")


def run(args, cwd, timeout=300):
    r = subprocess.run([sys.executable, GMX] + args, cwd=cwd, capture_output=True, text=True, timeout=timeout)
    return r


@pytest.fixture(scope="module")
def chain(tmp_path_factory):
    import stage_inputs as si
    d = str(tmp_path_factory.mktemp("chain"))
    si.write_system(d)
    for st in ("em", "nvt", "npt", "production"):
        si.write_mdp(os.path.join(d, st + ".mdp"), st)
    log = {}
    r = run(["grompp", "-f", "em.mdp", "-c", "conf.gro", "-r", "conf.gro", "-p", "topol.top", "-o", "em.tpr", "-maxwarn", "1000"], d)
    assert r.returncode == 0, r.stderr
    r = run(["mdrun", "-v", "-deffnm", "em", "-ntmpi", "1", "-ntomp", "4"], d)
    assert r.returncode == 0, r.stdout + r.stderr
    log["em"] = r.stdout
    r = run(["grompp", "-f", "nvt.mdp", "-c", "em.gro", "-r", "em.gro", "-p", "topol.top", "-o", "nvt.tpr", "-maxwarn", "1000"], d)
    assert r.returncode == 0, r.stderr
    r = run(["mdrun", "-v", "-deffnm", "nvt", "-ntmpi", "1", "-ntomp", "4"], d)
    assert r.returncode == 0, r.stdout + r.stderr
    log["nvt"] = r.stdout
    r = run(["grompp", "-f", "npt.mdp", "-c", "nvt.gro", "-r", "nvt.gro", "-t", "nvt.cpt", "-p", "topol.top", "-o", "npt.tpr",
             "-maxwarn", "1000"], d)
    assert r.returncode == 0, r.stderr
    r = run(["mdrun", "-v", "-deffnm", "npt", "-ntmpi", "1", "-ntomp", "4"], d)
    assert r.returncode == 0, r.stdout + r.stderr
    log["npt"] = r.stdout
    return d, log


def test_minimisation_lowers_energy_and_force(chain):
    d, log = chain
    assert "Steepest Descents" in log["em"]
    import re
    m = re.search(r"Potential Energy\s+=\s+(\S+)\nMaximum force\s+=\s+(\S+)\n\(started from Epot = (\S+), Fmax = (\S+)\)", log["em"])
    assert m, log["em"]
    e1, f1, e0, f0 = (float(m.group(i).rstrip(",")) for i in (1, 2, 3, 4))
    assert e1 < e0 and f1 < f0
    assert os.path.exists(os.path.join(d, "em.gro")) and os.path.exists(os.path.join(d, "em.log"))


def test_stage_outputs_exist_and_are_readable(chain):
    d, log = chain
    from mdx.io import gro, traj, energy, tpr
    for st, files in (("nvt", ["gro", "cpt", "log", "edr", "trr"]), ("npt", ["gro", "cpt", "log", "edr", "xtc"])):
        for ext in files:
            p = os.path.join(d, f"{st}.{ext}")
            assert os.path.exists(p) and os.path.getsize(p) > 0, p
    g = gro.read_gro(os.path.join(d, "nvt.gro"))
    assert g.v is not None and np.isfinite(g.x).all()
    fr = traj.read_trr(os.path.join(d, "nvt.trr"))
    assert [f["step"] for f in fr] == [0, 100, 200, 300]
    assert fr[0]["v"] is not None
    # final trr frame and final .gro agree to the .gro precision
    assert np.abs(fr[-1]["x"] - g.x).max() < 1e-3
    xf = traj.read_xtc(os.path.join(d, "npt.xtc"))
    assert [f["step"] for f in xf] == [0, 100, 200, 300]
    cp = tpr.load_checkpoint(os.path.join(d, "npt.cpt"))
    assert cp["step"] == 300
    assert np.abs(xf[-1]["x"] - cp["x"]).max() < 1e-3
    names, units, frames = energy.read_edr(os.path.join(d, "nvt.edr"))
    assert [f[0] for f in frames] == [0, 50, 100, 150, 200, 250, 300]
    with open(os.path.join(d, "npt.log")) as f:
        text = f.read()
    assert "Performance:" in text and "Pressure" in text


def test_nvt_temperature_and_constraints(chain):
    d, _ = chain
    from mdx.io import energy, gro
    names, units, frames = energy.read_edr(os.path.join(d, "nvt.edr"))
    T = np.array([f[2][names.index("Temperature")] for f in frames])
    # gen_vel at 300 K: the first frame sits at the target; the briefly minimised lattice then
    # releases potential energy and the thermostat (tau_t 0.1 ps) pulls the temperature back
    assert abs(T[0] - 300.0) < 8.0, T
    assert abs(T[-1] - 300.0) < 30.0 and T[-1] < T[1:4].max(), T
    g = gro.read_gro(os.path.join(d, "nvt.gro"))
    # rigid water geometry in the written frame (0.001 nm file precision)
    ow = np.array([i for i, (r, a) in enumerate(zip(g.resname, g.atomname)) if r == "SOL" and a == "OW"])
    doh = np.linalg.norm(g.x[ow] - g.x[ow + 1], axis=1)
    assert np.abs(doh - 0.09572).max() < 2.5e-3


def test_restart_from_checkpoint_appends(chain):
    d, _ = chain
    r = run(["grompp", "-f", "production.mdp", "-c", "npt.gro", "-t", "npt.cpt", "-p", "topol.top", "-o", "md.tpr"], d)
    assert r.returncode == 0, r.stderr
    r = run(["mdrun", "-deffnm", "md", "-nsteps", "200"], d)
    assert r.returncode == 0, r.stdout + r.stderr
    from mdx.io import traj, tpr
    assert tpr.load_checkpoint(os.path.join(d, "md.cpt"))["step"] == 200
    n1 = len(traj.read_xtc(os.path.join(d, "md.xtc")))
    assert n1 == 21
    r = run(["mdrun", "-deffnm", "md", "-cpi", "md.cpt"], d)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "continuing from step 200" in r.stdout
    with open(os.path.join(d, "md.log")) as f:
        assert "Restored the coupling and pull state from the checkpoint" in f.read()
    assert len(tpr.load_checkpoint(os.path.join(d, "md.cpt"))["extra"]["run_state"]) == 64
    fr = traj.read_xtc(os.path.join(d, "md.xtc"))
    steps = [f["step"] for f in fr]
    assert steps == list(range(0, 401, 10))               # no duplicate of the frame at the restart step
    # the recovery idiom: continue for "1 step" from a finished run leaves the final files
    r = run(["mdrun", "-deffnm", "md", "-cpi", "md.cpt", "-nsteps", "1"], d, timeout=30)
    assert r.returncode == 0 and os.path.exists(os.path.join(d, "md.gro"))


def test_smd_monitor_and_sigterm(chain):
    """Start the pull stage, tail PREFIX_pullx.xvg like the pipeline's monitor and stop the run
    with SIGTERM: mdrun has to leave .gro, .cpt and both pull files and exit within 30 s."""
    d, _ = chain
    import stage_inputs as si
    si.write_mdp(os.path.join(d, "pull.mdp"), "pull")
    r = run(["grompp", "-f", "pull.mdp", "-c", "npt.gro", "-t", "npt.cpt", "-p", "topol.top", "-n", "index.ndx", "-o", "smd.tpr",
             "-maxwarn", "1000"], d)
    assert r.returncode == 0, r.stderr
    assert "nstcheckpoint" in r.stderr
    env = dict(os.environ, OMP_NUM_THREADS="4")
    p = subprocess.Popen([sys.executable, GMX, "mdrun", "-v", "-deffnm", "smd", "-ntmpi", "1", "-ntomp", "4"], cwd=d, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, bufsize=1)
    pullx = os.path.join(d, "smd_pullx.xvg")
    rows, pos, t0 = [], 0, time.time()
    while time.time() - t0 < 120 and len(rows) < 40:
        if os.path.exists(pullx):
            with open(pullx) as f:
                f.seek(pos)
                for line in f.readlines():
                    if line.strip() and line[0] not in "#@":
                        parts = line.split()
                        rows.append((float(parts[0]), float(parts[1])))
                pos = f.tell()
        if p.poll() is not None:
            break
        time.sleep(0.2)
    assert len(rows) >= 40, (len(rows), p.poll())
    p.send_signal(signal.SIGTERM)
    try:
        outtxt, _ = p.communicate(timeout=30)
    except subprocess.TimeoutExpired:
        p.kill()
        pytest.fail("mdrun did not stop within 30 s of SIGTERM")
    assert p.returncode == 0, outtxt
    assert "step " in outtxt
    for f in ("smd.gro", "smd.cpt", "smd_pullx.xvg", "smd_pullf.xvg", "smd.log"):
        assert os.path.exists(os.path.join(d, f)), f
    from mdx.io import xvg
    X, _ = xvg.read_xvg(pullx)
    F, _ = xvg.read_xvg(os.path.join(d, "smd_pullf.xvg"))
    # samples every 50 steps of 1 fs; constant force of -k
    assert np.allclose(np.diff(X[:, 0]), 0.05, atol=1e-6)
    assert np.allclose(F[:, 1], -500.0)
    # attraction: the distance shrinks on average
    assert X[-1, 1] < X[0, 1]
