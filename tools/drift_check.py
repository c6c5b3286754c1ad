#!/usr/bin/env python3
"""NVE energy drift of the S2 pseudo-complex against the pair-list radius (nstlist 20, dt 2 fs) and on the dual pair list
(outer list 100 steps, pruned every 10): checks that the radii from the verlet-buffer-tolerance estimate keep the
measured drift under the tolerance (0.005 kJ/mol/ps per atom)."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=float(os.environ.get("DRIFT_BOX", "12.5")))
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstlist=10, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2)
e.step(1500)
x, v, box = e.get_state(); e.close()
s1 = copy.copy(s); s1.x = x.astype(np.float64); s1.v = v.astype(np.float64); s1.box = box.astype(np.float64)
e = eng.Engine(s1, rc=1.0, dt=0.002, nstlist=10, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.1, 0.1], nsttcouple=10)
e.step(3000)
x, v, box = e.get_state(); e.close()
s2 = copy.copy(s); s2.x = x.astype(np.float64); s2.v = v.astype(np.float64); s2.box = box.astype(np.float64)
nsteps = int(os.environ.get("DRIFT_STEPS", "10000"))
auto = None
for rl, nstlist, nstprune in ((None, 20, 0), (None, 100, 10), (None, 120, 12), (1.02, 20, 0), (1.05, 20, 0), (1.127, 20, 0), (1.2, 20, 0)):
    e = eng.Engine(s2, rc=1.0, rlist=rl, dt=0.002, nstlist=nstlist, nstprune=nstprune, nstcalcenergy=10, nstcomm=100)
    if rl is None: auto = e.rlist
    e.step(nsteps)
    fr = e.pop_energy_frames()
    t = fr[:, 1]; etot = fr[:, 2 + eng.E["POT"]] + fr[:, 2 + eng.E["KIN"]]
    slope = np.polyfit(t, etot, 1)[0] / s.natoms
    print("nstlist %3d nstprune %2d rlist %.3f inner %s%s: drift %+.5f kJ/mol/ps per atom   (E_tot %.1f -> %.1f, T %.1f K)" % (
        nstlist, nstprune, e.rlist, e.rlist_inner, " (estimate)" if rl is None else "", slope, etot[0], etot[-1], fr[-1, 2 + eng.E["TEMP"]] if "TEMP" in eng.E else -1), flush=True)
    e.close()
