#!/usr/bin/env python3
"""200 ps of the bench workload with the production settings (V-rescale on two groups, Parrinello-Rahman, DispCorr EnerPres,
pair list renewed every 20 steps): temperature, pressure, density and the drift of the conserved quantity, plus the
constraint geometry at the end. A sanity record of the step as a whole, not a parity test."""
import os
import sys
import copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mdx  # noqa: F401
from mdx import synth, engine as eng

s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstlist=10, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2)
e.step(2000); x, v, box = e.get_state(); e.close()
s1 = copy.copy(s); s1.x = x.astype(np.float64); s1.v = v.astype(np.float64); s1.box = box.astype(np.float64)
e = eng.Engine(s1, rc=1.0, dt=0.002, nstlist=20, nstcalcenergy=10, tc_groups=tc, ref_t=[300, 300], tau_t=[0.1, 0.1], nsttcouple=10,
               pcoupl=2, tau_p=2.0, ref_p=1.0, compressibility=4.5e-5, nstpcouple=10, dispcorr=2, nstcomm=100)
nsteps = int(os.environ.get("LONG_STEPS", "100000"))
e.step(5000); e.pop_energy_frames()            # settle into the barostat
frames = []
for _ in range(nsteps // 5000):
    e.step(5000)
    frames.append(e.pop_energy_frames())
fr = np.concatenate(frames)
t = fr[:, 1]
T, P, V, C = (fr[:, 2 + eng.E[k]] for k in ("TEMP", "PRES", "VOLUME", "CONSERVED"))
mass = s.mass.sum() * 1.66054e-27
print("steps %d (%.0f ps), %d energy frames" % (nsteps, nsteps * 0.002, len(fr)))
print("T   mean %.2f K  sd %.2f" % (T.mean(), T.std()))
print("P   mean %.1f bar  sd %.0f" % (P.mean(), P.std()))
print("density mean %.1f kg/m^3  (volume %.2f -> %.2f nm^3)" % (mass / (V.mean() * 1e-27), V[0], V[-1]))
print("conserved quantity drift %+.5f kJ/mol/ps per atom" % (np.polyfit(t, C, 1)[0] / s.natoms))
x, v, box = e.get_state(); e.close()
x = x.astype(np.float64)
w = s.settles
print("water O-H max relative deviation %.2e" % np.abs(np.linalg.norm(x[w[:, 0]] - x[w[:, 1]], axis=1) / s.settle_doh - 1).max())
c = s.constraints
print("h-bond constraints max relative deviation %.2e" % np.abs(np.linalg.norm(x[c[:, 0]] - x[c[:, 1]], axis=1) / s.constraint_len - 1).max())
print("finite: %s" % bool(np.isfinite(x).all() and np.isfinite(v).all()))
