#!/usr/bin/env python3
"""Charge spreading alone (stage timer) for the kernel selected by MDX_SPREAD (col | chunk)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=20, nstcalcenergy=0, tc_groups=["System"], ref_t=[300], tau_t=[0.1], nsttcouple=10)
e.step(200); e.synchronize()
print(os.environ.get("MDX_SPREAD", "chunk"), "spread alone %.1f us, gather %.1f us, whole chain %.1f us" % (e.debug_time_stage(6, 50) * 1e3, e.debug_time_stage(7, 50) * 1e3, e.debug_time_stage(2, 50) * 1e3))
