#!/usr/bin/env python3
"""Launch overhead of the force-only nonbonded kernel: the launch as HIP events see it against the span from its first
wave's start to its last wave's end (in-kernel stamps; library built with -DMDX_STAMPS).
Usage (GPU box): MDX_LIB=molecular-dynamics_amd/libmdx_ts.so python tools/nb_overhead.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=100, nstprune=10, tc_groups=["Protein", "Non-Protein"], ref_t=[300, 300], tau_t=[0.1, 0.1], nstcalcenergy=0)
e.step(315); e.synchronize()
for n in (1, 1, 1, 5):
    e.debug_stamps(reset=True)
    tev = e.debug_time_nonbonded(False, n) * 1e3
    t = e.debug_stamps()
    print("launches %d: event %.1f us per launch; stamp of the last launch: first wave start -> last wave end %.1f us" % (n, tev, (t[1][1] - t[0][1]) / 100.0))
e.close()
