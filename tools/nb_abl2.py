#!/usr/bin/env python3
"""Nonbonded kernel with arithmetic removed at compile time (diagnostic; forces are wrong in those modes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstlist=20, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2, rlist=1.095)
e.step(500); e.synchronize()
print(e.list_stats())
e.debug_set(3, 1); e.debug_set(4, 0)
for rep in range(2):
    for bits, label in ((0, "full, tabulated correction"), (0x400000, "full, rational correction"), (0x100000, "no Ewald correction polynomial"), (0x200000, "no j-force reduction")):
        e.debug_set(2, bits)
        print("%-32s %.1f us" % (label, e.debug_time_nonbonded(False, 30) * 1e3), flush=True)
e.debug_set(2, 0)
