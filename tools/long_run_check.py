#!/usr/bin/env python3
"""A long production-settings run of the bench system (default 100 000 steps = 200 ps, V-rescale on two groups,
Parrinello-Rahman, DispCorr EnerPres, dual pair list, energies every 10 steps): averages and drifts of temperature,
pressure, density and the conserved quantity over the second half, and the rigid-water geometry at the end.
Usage (GPU box): python tools/long_run_check.py [nsteps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstlist=10, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2)
e.step(2000); x, v, box = e.get_state(); e.close()
import copy
s1 = copy.copy(s); s1.x = x.astype(np.float64); s1.v = v.astype(np.float64); s1.box = box.astype(np.float64)
e = eng.Engine(s1, rc=1.0, dt=0.002, nstlist=100, nstprune=10, nstcalcenergy=10, tc_groups=tc, ref_t=[300, 300], tau_t=[0.1, 0.1], nsttcouple=10,
               pcoupl=2, tau_p=2.0, ref_p=1.0, compressibility=4.5e-5, nstpcouple=10, dispcorr=2, nstcomm=100)
t0 = time.time()
rows = []
done = 0
while done < n:
    m = min(20000, n - done)
    e.step(m); done += m
    rows.append(e.pop_energy_frames())
fr = np.concatenate(rows)
wall = time.time() - t0
E = eng.E
half = fr[len(fr) // 2:]
t = half[:, 1]
def stat(name):
    y = half[:, 2 + E[name]]
    return y.mean(), y.std(), np.polyfit(t, y, 1)[0]
mass = float(np.sum(s.mass))
print("%d steps in %.1f s (%.1f us/step, energy frames popped every 20000 steps)" % (n, wall, wall / n * 1e6))
for name, unit in (("TEMP", "K"), ("PRES", "bar"), ("VOLUME", "nm^3"), ("POT", "kJ/mol"), ("CONSERVED", "kJ/mol")):
    m_, sd, sl = stat(name)
    print("%-10s mean %14.4f  std %12.4f  slope %+.4e %s/ps" % (name, m_, sd, sl, unit))
vol = stat("VOLUME")[0]
print("density    %.2f kg/m^3" % (mass * 1.66053907 / vol))
print("conserved-quantity drift per atom: %+.2e kJ/mol/ps" % (stat("CONSERVED")[2] / s.natoms))
x, v, box = e.get_state()
x = x.astype(np.float64)
w = np.asarray(s.settles)
doh = np.linalg.norm(x[w[:, 0]] - x[w[:, 1]], axis=1); dhh = np.linalg.norm(x[w[:, 1]] - x[w[:, 2]], axis=1)
print("water geometry: |d_OH / d0 - 1| max %.1e, |d_HH / d0 - 1| max %.1e; finite: %s" % (np.abs(doh / s.settle_doh - 1).max(), np.abs(dhh / s.settle_dhh - 1).max(), bool(np.isfinite(x).all())))
e.close()
