#!/usr/bin/env python3
"""Timing of the pair search with parts of the list builder switched off (diagnostic; lists are wrong in those modes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2)
e.step(500); e.synchronize()
for bits, label in ((0, "full"), (8192, "no sure-accept shortcut"), (16384, "no exclusion lookup"), (512, "no atom-level tests"), (1024, "no cluster-level work"), (4096, "candidate search only"),
                    (2048, "central shift only"), (2048 + 4096, "central shift, candidate search only")):
    e.debug_set(2, bits)
    try:
        ts = e.debug_time_stage(1, 20) * 1e3
        st = e.list_stats()
        print("%-40s %.1f us   nb force %.1f us   %s" % (label, ts, e.debug_time_nonbonded(False, 20) * 1e3 if bits in (0, 8192) else 0.0, st), flush=True)
    except Exception as ex:
        print(label, "failed:", ex, flush=True)
e.debug_set(2, 0)
