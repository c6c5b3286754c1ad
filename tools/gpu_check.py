#!/usr/bin/env python3
"""First-contact diagnostic on the GPU box: force parity per term vs the float64 oracle,
short NVE run, rough timing. Writes gpurun_out/gpu_check.log."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mdx
from mdx import synth, engine as eng
from oracle import mdref as R

def relrms(a, b):
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))

def parity(name, s, rc=1.0, **kw):
    s.x = s.x.astype(np.float32).astype(np.float64)   # both sides see the same float32 inputs
    print(f"== {name}: {s.natoms} atoms", flush=True)
    e = eng.Engine(s, rc=rc, nstcalcenergy=1, **kw)
    beta, grid = e.beta, e.grid
    print("   beta", beta, "grid", grid, "rlist", e.rlist, flush=True)
    # nonbonded
    f, en = e.compute(eng.F_NONBONDED)
    print("   list", e.list_stats(), flush=True)
    o = R.nonbonded(s.x, s.charge, s.sigma, s.eps, s.box, rc, beta, s.excl_start, s.excl_list, flags=1, nthreads=8)
    print(f"   NB   force relrms {relrms(f, o['f']):.3e}  E_LJ {en['LJ']:.6f} vs {o['e_lj']:.6f}  E_C {en['COUL_SR']:.6f} vs {o['e_coul_sr']:.6f} npairs {o['npairs']}", flush=True)
    v, p = e.virial()
    print("   NB   vir diag", np.diag(v), "oracle", np.diag(o['vir']), flush=True)
    # exclusion correction
    f, en = e.compute(eng.F_EXCL)
    o2 = R.nonbonded(s.x, s.charge, s.sigma, s.eps, s.box, rc, beta, s.excl_start, s.excl_list, flags=2, nthreads=1)
    print(f"   EXCL force relrms {relrms(f, o2['f']):.3e}  E {en['COUL_EXCL']:.6f} vs {o2['e_excl']:.6f}", flush=True)
    # pme
    f, en = e.compute(eng.F_PME)
    Ep, Fp, vp = R.pme_recip(s.x, s.charge, s.box, beta, grid)
    print(f"   PME  force relrms {relrms(f, Fp):.3e}  E {en['COUL_RECIP']:.6f} vs {Ep:.6f}", flush=True)
    v, p = e.virial()
    print("   PME  vir diag", np.diag(v), "oracle", np.diag(vp), flush=True)
    # bonded
    if len(s.bonds) or len(s.angles):
        f, en = e.compute(eng.F_BONDED)
        eb, fb, _ = R.bonds(s.x, s.bonds, s.bond_b0, s.bond_kb)
        ea, fa, _ = R.angles(s.x, s.angles, s.angle_th0, s.angle_k)
        ed, fd, _ = R.dihedrals(s.x, s.dihedrals, s.dih_phi0, s.dih_k, s.dih_mult)
        el, ec, fp14, _ = R.pairs14(s.x, s.pairs14, s.pair_c6, s.pair_c12, s.pair_qq)
        print(f"   BOND force relrms {relrms(f, fb+fa+fd+fp14):.3e} E {en['BOND']:.4f}/{eb:.4f} {en['ANGLE']:.4f}/{ea:.4f} {en['DIH']:.4f}/{ed:.4f} {en['LJ14']:.4f}/{el:.4f} {en['COUL14']:.4f}/{ec:.4f}", flush=True)
    # all
    f, en = e.compute(eng.F_ALL & ~eng.F_PULL & ~eng.F_POSRES)
    ref = R.all_forces(s, rc, beta, grid, nthreads=8)
    print(f"   ALL  force relrms {relrms(f, ref['f']):.3e}  sum f {np.abs(f.sum(0)).max():.3e}  frms {np.sqrt((f**2).mean()):.2f}", flush=True)
    return e

if __name__ == "__main__":
    t0 = time.time()
    s = synth.water_box(d=4.0)
    e = parity("water-dodeca-4.0", s)
    # short NVE: energy conservation
    e2 = eng.Engine(s, rc=1.0, nstcalcenergy=10, nstcomm=0, dt=0.001)
    e2.step(2001)
    fr = e2.pop_energy_frames()
    tot = fr[:, 2 + eng.E["POT"]] + fr[:, 2 + eng.E["KIN"]]
    print("NVE 2 ps dt=1fs: E_tot first/last", tot[0], tot[-1], "drift/atom/ps", (tot[-1] - tot[0]) / s.natoms / 2.0,
          "T", fr[0, 2 + eng.E["TEMP"]], fr[-1, 2 + eng.E["TEMP"]], flush=True)
    print("   Epot", fr[::20, 2 + eng.E["POT"]], flush=True)
    x, v, box = e2.get_state()
    d1 = np.linalg.norm(x[0::3] - x[1::3], axis=1); d3 = np.linalg.norm(x[1::3] - x[2::3], axis=1)
    print("   OH min/max", d1.min(), d1.max(), "HH", d3.min(), d3.max(), flush=True)
    e.close(); e2.close()
    s = synth.pseudo_complex(d=7.0, protein_radius=1.2, n_active=60)
    e = parity("pseudo-complex-7.0", s)
    e.close()
    # timing on the 130k water box
    s = synth.water_box(d=12.5)
    e = eng.Engine(s, rc=1.0, nstcalcenergy=100, nstlist=10, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1])
    e.step(200); e.synchronize()
    print("list", e.list_stats(), flush=True)
    t = time.time(); e.step(1000); e.synchronize(); dt = time.time() - t
    print(f"water-130k: {s.natoms} atoms  {dt/1000*1e3:.3f} ms/step  {0.002*1000/dt*86400/1000:.1f} ns/day", flush=True)
    en = e.energies(); print("T", en["TEMP"], "P", en["PRES"], "Epot", en["POT"], flush=True)
    e.enable_kernel_timing(True); e.step(200); ms, nl = e.kernel_timing(); print("nb kernel avg ms", ms / max(nl, 1), nl, flush=True)
    print("total wall", time.time() - t0)
