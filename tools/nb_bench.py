#!/usr/bin/env python3
"""Micro-benchmark of the nonbonded kernel variants on one pair list (interleaved rounds, one process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
from oracle import mdref as R
s = synth.water_box(d=12.5)
e = eng.Engine(s, rc=1.0, nstcalcenergy=100, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1])
e.step(300); e.synchronize()
f, en = e.compute(eng.F_NONBONDED)
st = e.list_stats(); print(st)
flops = 66.0 * st["pairs_with_lj"] + 54.0 * (st["pairs_within_rc"] - st["pairs_with_lj"])
res = {1: [], 2: []}
for rnd in range(5):
    for v in (1, 2):
        e.debug_set(1, v)
        res[v].append(e.debug_time_nonbonded(False, 40))
for v in (1, 2):
    t = np.array(res[v]); print(f"variant {v}: median {np.median(t)*1e3:.1f} us  min {t.min()*1e3:.1f} us  -> {flops/np.median(t)/1e9:.2f} TFLOP/s algorithmic")
# parity of each variant on a smaller box
s2 = synth.water_box(d=4.0); s2.x = s2.x.astype(np.float32).astype(np.float64)
e2 = eng.Engine(s2, rc=1.0, nstcalcenergy=1)
o = R.nonbonded(s2.x, s2.charge, s2.sigma, s2.eps, s2.box, 1.0, e2.beta, s2.excl_start, s2.excl_list, flags=1, nthreads=8)
for v in (1, 2):
    e2.debug_set(1, v)
    f, en = e2.compute(eng.F_NONBONDED)
    print(f"variant {v} energy-kernel force relrms {np.sqrt(((f-o['f'])**2).sum()/(o['f']**2).sum()):.3e}")
    # force-only kernel: one step of dynamics uses it; compare via a zero-velocity single step displacement
