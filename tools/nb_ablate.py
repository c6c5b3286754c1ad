#!/usr/bin/env python3
"""Nonbonded kernel alone with parts switched off (diagnostic; forces are wrong in those modes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstlist=20, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2)
e.step(500); e.synchronize()
print(e.list_stats())
for var in (33, 34):
    e.debug_set(1, var)
    for bits, label in ((0, "full"), (0x20000, "no j-force atomics"), (0x40000, "no i-force atomics"), (0x60000, "no atomics")):
        e.debug_set(2, bits)
        print("variant %d %-22s %.1f us" % (var, label, e.debug_time_nonbonded(False, 30) * 1e3), flush=True)
e.debug_set(2, 0); e.debug_set(1, 33)
