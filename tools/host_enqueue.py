#!/usr/bin/env python3
"""Host enqueue time per MD step against GPU time per step (S2 system): is the stepping thread ahead of the GPU?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstlist=20, nstcalcenergy=10, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=10)
e.step(1000); e.synchronize()
for n in (5, 10, 20, 40, 80, 160):
    hs, gs = [], []
    for rep in range(5):
        e.synchronize(); t0 = time.perf_counter()
        e.step(n); t1 = time.perf_counter()
        e.synchronize(); t2 = time.perf_counter()
        hs.append((t1 - t0) / n * 1e6); gs.append((t2 - t0) / n * 1e6)
    print("steps %4d: host enqueue %.1f us/step, until done %.1f us/step" % (n, min(hs), min(gs)), flush=True)
