#!/usr/bin/env python3
"""Energy/virial nonbonded kernel alone: tabulated (default) against the analytic erfc variant, and the energies both give."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstlist=20, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2, rlist=1.095)
e.step(500); e.synchronize()
for bits, label in ((0, "tabulated"), (0x400000, "analytic")):
    e.debug_set(2, bits)
    f, en = e.compute()
    print("%-10s %.1f us   LJ %.4f  Coul-SR %.4f  excl %.4f" % (label, e.debug_time_nonbonded(True, 30) * 1e3, en["LJ"], en["COUL_SR"], en.get("COUL_EXCL", float("nan"))), flush=True)
e.debug_set(2, 0)
