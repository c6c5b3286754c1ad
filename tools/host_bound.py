#!/usr/bin/env python3
"""How long does the host take to enqueue steps vs. how long does the GPU take to run them?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2)
e.step(1500); e.synchronize()
for n in (200, 1000):
    t0 = time.perf_counter(); e.step(n); t1 = time.perf_counter(); e.synchronize(); t2 = time.perf_counter()
    print(f"{n} steps: enqueue {1e6*(t1-t0)/n:.1f} us/step, total {1e6*(t2-t0)/n:.1f} us/step")
