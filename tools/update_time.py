#!/usr/bin/env python3
"""The fused update alone on the bench system: average period of back-to-back launches (mdx_debug_time_stage 8: with the
next step's sorted coordinates, 9: without), and leap-frog + constraints as separate launches (4).
Usage (GPU box): python tools/update_time.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=100, nstprune=10, tc_groups=["Protein", "Non-Protein"], ref_t=[300, 300], tau_t=[0.1, 0.1], nstcalcenergy=0)
e.step(200); e.synchronize()
for stage, label in ((8, "fused update + sorted coordinates"), (9, "fused update"), (4, "leap-frog + constraint launches"), (5, "prepare")):
    ms = ctypes.c_double(0)
    for rep in range(2):
        e._call("mdx_debug_time_stage", stage, 200, ctypes.byref(ms))
    print("%-36s %.2f us" % (label, ms.value * 1e3))
e.close()
