#!/usr/bin/env python3
"""Runs the force-only nonbonded kernel a few times on the 140k-atom water box (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdx
from mdx import synth, engine as eng
s = synth.water_box(d=12.5)
e = eng.Engine(s, rc=1.0, nstcalcenergy=100, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1])
e.step(300); e.synchronize()
e.compute(eng.F_NONBONDED)
print(e.debug_time_nonbonded(False, 5))
