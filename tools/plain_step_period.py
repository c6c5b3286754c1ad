#!/usr/bin/env python3
"""Period of a plain step on the bench workload: a run without energy steps, pressure coupling or output (thermostat every
1000 steps), against the production settings; the difference per step is what energy, coupling and search steps add."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
def run(label, **kw):
    e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=100, nstprune=10, tc_groups=tc, ref_t=[300, 300], tau_t=[0.1, 0.1], **kw)
    e.step(3000); e.synchronize()
    t0 = time.perf_counter(); e.step(4000); e.synchronize(); dt = (time.perf_counter() - t0) / 4000 * 1e6
    print("%-64s %.1f us/step" % (label, dt)); e.close()
run("plain steps only (no energies, no barostat, thermostat every 1000)", nstcalcenergy=0, nsttcouple=1000)
run("+ thermostat every 10 steps", nstcalcenergy=0, nsttcouple=10)
run("+ energies every 10 steps", nstcalcenergy=10, nsttcouple=10)
run("+ Parrinello-Rahman every 10 steps, dispersion correction", nstcalcenergy=10, nsttcouple=10, pcoupl=2, tau_p=2.0, ref_p=1.0, compressibility=4.5e-5, nstpcouple=10, dispcorr=2)
