#!/usr/bin/env python3
"""Runs the force-only nonbonded kernel a few times on the bench workload (S2 pseudo-complex, dual pair list as in bench.py:
outer list 100 steps, pruned every 10) for rocprofv3 --pmc."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstlist=100, nstprune=10, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2)
for kv in filter(None, os.environ.get("MDX_DEBUG", "").split(",")):     # diagnostic switches "key=value,..."
    k, v = kv.split("="); e.debug_set(int(k), int(v, 0))
e.step(215); e.synchronize()
print("rlist", e.rlist, e.list_stats())
print(e.debug_time_nonbonded(bool(os.environ.get("MDX_NB_ENERGY")), 20))      # MDX_NB_ENERGY=1: the energy kernel
