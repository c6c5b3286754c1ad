#!/usr/bin/env python3
"""Per-atom forces and energy terms of this engine against a real GROMACS, when one is installed.

The pipeline's arithmetic lives in `gmx mdrun` (/root/reference/envs/environment.yml:24, `gromacs>=2021`), which is neither
vendored nor installed where this repository is built, so the parity tests compare the HIP path with an independent
float64 restatement. This script is the direct comparison for a machine that has GROMACS:

  1. the bench system (`--box 12.5`: 138 792 atoms; `--box 6.0` for a quick run) is written as conf.gro + topol.top with its
     own force-field directory (io/top.write_top_tree), the way the pipeline hands a system to `gmx grompp`;
  2. `gmx grompp` + `gmx mdrun -rerun conf.gro -nb cpu -pme cpu -bonded cpu -ntmpi 1 -ntomp 4` with production.mdp's
     interaction settings (/root/reference/mdp/production.mdp:21-29, 46) and nstcalcenergy = nstenergy = nstfout = 1
     evaluate that one configuration on the reference's CPU path;
  3. forces are read with `gmx dump -f rerun.trr`, energies with `gmx energy -f rerun.edr` (GROMACS' own readers, so no
     assumption about binary layouts enters);
  4. the same configuration goes through mdx_compute_forces on the GPU and the two are compared: force relative RMS and the
     energy terms, tolerance 1e-5 (`--tol`).

Without a GROMACS binary it prints `reference unavailable` and exits 0. Needs a GPU for step 4. Not part of the test suite:
no machine of this build has GROMACS."""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def find_gromacs():
    """A real GROMACS (not this repository's bin/gmx shim)."""
    shim = os.path.realpath(os.path.join(ROOT, "bin", "gmx"))
    for name in ("gmx", "gmx_mpi", "gmx_d"):
        p = shutil.which(name)
        if not p or os.path.realpath(p) == shim:
            continue
        try:
            with open(p, "rb") as f:
                if b"molecular-dynamics_amd" in f.read(512):
                    continue
        except OSError:
            continue
        return p
    return None


RERUN_MDP = """integrator = md
nsteps = 0
dt = 0.002
nstcalcenergy = 1
nstenergy = 1
nstfout = 1
nstxout = 1
continuation = yes
constraints = h-bonds
constraint_algorithm = lincs
cutoff-scheme = Verlet
nstlist = 10
rcoulomb = 1.0
rvdw = 1.0
coulombtype = PME
pme_order = 4
fourierspacing = 0.16
ewald_rtol = 1e-5
tcoupl = no
pcoupl = no
pbc = xyz
DispCorr = EnerPres
"""

# gmx energy term -> sum of this engine's terms (the excluded-pair Ewald correction is part of GROMACS' short-range
# Coulomb kernel, the self energy of its reciprocal term; only their sum is convention-free)
TERMS = [("Bond", ("BOND",)), ("Angle", ("ANGLE",)), ("Proper-Dih.", ("DIH",)), ("LJ-14", ("LJ14",)), ("Coulomb-14", ("COUL14",)),
         ("LJ-(SR)", ("LJ",)), ("Disper.-corr.", ("DISPCORR",)), ("Potential", ("POT",))]
COULOMB = (("Coulomb-(SR)", "Coul.-recip."), ("COUL_SR", "COUL_EXCL", "COUL_RECIP"))


def run(cmd, cwd, stdin=None):
    r = subprocess.run(cmd, cwd=cwd, input=stdin, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("%s failed:\n%s" % (" ".join(cmd), (r.stderr or r.stdout)[-2000:]))
    return r.stdout


def reference_forces(gmx, d, natoms):
    import numpy as np
    txt = run([gmx, "dump", "-f", "rerun.trr"], d)
    f = np.full((natoms, 3), np.nan)
    for m in re.finditer(r"f\[\s*(\d+)\]=\{\s*([-+0-9.eE]+),\s*([-+0-9.eE]+),\s*([-+0-9.eE]+)\}", txt):
        f[int(m.group(1))] = [float(m.group(2)), float(m.group(3)), float(m.group(4))]
    if np.isnan(f).any():
        raise RuntimeError("gmx dump returned forces for %d of %d atoms" % (int((~np.isnan(f[:, 0])).sum()), natoms))
    return f


def reference_energies(gmx, d):
    from mdx.io import xvg
    names = [t for t, _ in TERMS] + list(COULOMB[0])
    out = {}
    for nm in names:
        try:
            run([gmx, "energy", "-f", "rerun.edr", "-o", "e.xvg"], d, stdin=nm + "\n\n")
        except RuntimeError:
            continue            # a term the system does not have (no dihedrals in a water box, ...)
        data, _ = xvg.read_xvg(os.path.join(d, "e.xvg"))          # rows of (time, value)
        out[nm] = float(data[0, 1]) if data.size else float("nan")
        os.remove(os.path.join(d, "e.xvg"))
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--box", type=float, default=12.5, help="dodecahedron image distance of the generated system in nm")
    ap.add_argument("--tol", type=float, default=1e-5)
    ap.add_argument("--keep", action="store_true", help="keep the working directory")
    args = ap.parse_args()
    gmx = find_gromacs()
    if gmx is None:
        print("reference unavailable")
        return 0
    import numpy as np
    import mdx  # noqa: F401
    from mdx import synth, engine as eng
    from mdx.io import gro, top
    s = synth.pseudo_complex(d=args.box)
    d = tempfile.mkdtemp(prefix="mdx_gmx_compare_")
    try:
        top.write_top_tree(d, s)
        gro.write_system_gro(os.path.join(d, "conf.gro"), s, x=s.x, v=np.zeros_like(s.x), box=s.box)
        with open(os.path.join(d, "rerun.mdp"), "w") as f:
            f.write(RERUN_MDP)
        run([gmx, "grompp", "-f", "rerun.mdp", "-c", "conf.gro", "-p", "topol.top", "-o", "rerun.tpr", "-maxwarn", "10"], d)
        run([gmx, "mdrun", "-s", "rerun.tpr", "-rerun", "conf.gro", "-deffnm", "rerun", "-nb", "cpu", "-pme", "cpu", "-bonded", "cpu",
             "-ntmpi", "1", "-ntomp", "4"], d)
        f_ref = reference_forces(gmx, d, s.natoms)
        e_ref = reference_energies(gmx, d)
        # the .gro file holds three decimals: evaluate what GROMACS evaluated
        fr = gro.read_gro(os.path.join(d, "conf.gro"))
        s.x = np.asarray(fr.x, dtype=s.x.dtype)
        log = open(os.path.join(d, "rerun.log")).read()
        m = re.search(r"Using a Fourier grid of (\d+)x(\d+)x(\d+)", log) or re.search(r"fourier-nx\s*=\s*(\d+)[\s\S]*?fourier-ny\s*=\s*(\d+)[\s\S]*?fourier-nz\s*=\s*(\d+)", log)
        grid = tuple(int(g) for g in m.groups()) if m else None
        e = eng.Engine(s, rc=1.0, nstcalcenergy=1, dispcorr=2, **({"grid": grid} if grid else {}))
        f, en = e.compute()
        e.close()
        bad = 0
        rel = float(np.sqrt(((f - f_ref) ** 2).sum() / (f_ref ** 2).sum()))
        worst = int(np.argmax(np.abs(f - f_ref).max(axis=1)))
        print("atoms %d, PME grid %s" % (s.natoms, grid))
        print("forces: relative RMS difference %.3e (largest per-atom difference %.3e kJ/mol/nm at atom %d)  %s" % (
            rel, float(np.abs(f - f_ref).max()), worst, "ok" if rel <= args.tol else "ABOVE TOLERANCE"))
        bad += rel > args.tol
        scale = max(abs(v) for v in e_ref.values()) if e_ref else 1.0
        rows = [(t, sum(en[k] for k in ours)) for t, ours in TERMS if t in e_ref]
        rows = [(t, e_ref[t], v) for t, v in rows]
        if all(t in e_ref for t in COULOMB[0]):
            rows.append(("Coulomb (SR + recip.)", sum(e_ref[t] for t in COULOMB[0]), sum(en[k] for k in COULOMB[1])))
        for name, a, b in rows:
            ok = abs(a - b) <= args.tol * max(abs(a), 1e-3 * scale)
            bad += not ok
            print("%-24s gmx %16.4f   mdx %16.4f   diff %10.3e  %s" % (name, a, b, b - a, "ok" if ok else "ABOVE TOLERANCE"))
        return 1 if bad else 0
    finally:
        if args.keep:
            print("working directory kept:", d)
        else:
            shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
