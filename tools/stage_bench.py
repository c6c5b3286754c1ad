#!/usr/bin/env python3
"""Per-stage timings (each stage alone on the GPU) of the S2 pseudo-complex step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2)
e.step(1500); e.synchronize()
names = {1: "search", 2: "pme chain", 3: "bonded+excl", 4: "update+constraints", 5: "prepare", 6: "spread only"}
print("nonbonded force-only: %.1f us" % (e.debug_time_nonbonded(False, 30) * 1e3))
print("nonbonded energy+virial: %.1f us" % (e.debug_time_nonbonded(True, 30) * 1e3))
for st in (1, 2, 6, 3, 5, 4):
    print("%-20s %.1f us" % (names[st], e.debug_time_stage(st, 20) * 1e3))
for bits, label in ((1, "spread, no flush"), (2, "spread, no LDS adds"), (3, "spread, neither"), (4, "spread, global atomics only")):
    e.debug_set(2, bits)
    print("%-28s %.1f us" % (label, e.debug_time_stage(6, 20) * 1e3))
e.debug_set(2, 0)
