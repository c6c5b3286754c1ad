#!/usr/bin/env python3
"""The force-only nonbonded kernel alone on the bench system's pruned pair list: average of back-to-back launches
(mdx_debug_time_nonbonded). MDX_DEBUG="key=value,..." applies engine switches first. Usage (GPU box): python tools/nb_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=100, nstprune=10, tc_groups=["Protein", "Non-Protein"], ref_t=[300, 300], tau_t=[0.1, 0.1], nstcalcenergy=0)
for kv in filter(None, os.environ.get("MDX_DEBUG", "").split(",")):
    e.debug_set(int(kv.split("=")[0]), int(kv.split("=")[1], 0))
if os.environ.get("NB_TIME_NOSTEP"):
    e.compute()          # the outer list of the start configuration, no dynamics
else:
    e.step(315); e.synchronize()
for rep in range(3):
    print("force kernel alone: %.2f us" % (e.debug_time_nonbonded(False, 200) * 1e3))
e.close()
