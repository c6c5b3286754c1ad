#!/usr/bin/env python3
"""Force-kernel alone time on the bench workload's pair list for values of the diagnostic bit field (mdx_debug_set key 2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
bits = [int(a, 0) for a in sys.argv[1:]] or [0]
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstlist=20, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2)
e.step(1500); e.synchronize()
for rep in range(2):
    for b in bits:
        e.debug_set(2, b)
        print("bits 0x%x: force kernel alone %.1f us" % (b, e.debug_time_nonbonded(False, 40) * 1e3))
e.debug_set(2, 0)
