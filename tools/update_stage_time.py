#!/usr/bin/env python3
"""The update kernels alone on the bench system, back to back on the main stream (HIP events): leap-frog + constraints
(two launches), prepare, and the fused update with and without the sorted coordinates of the next step."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdx  # noqa: E402,F401
from mdx import synth, engine as eng  # noqa: E402
s = synth.pseudo_complex(d=12.5)
e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=20, nstcalcenergy=0, tc_groups=["Protein", "Non-Protein"], ref_t=[300, 300], tau_t=[0.1, 0.1])
e.step(200); e.synchronize()
for stage, name in ((4, "leap-frog + constraints"), (5, "prepare"), (8, "fused update + sorted coordinates"), (9, "fused update"),
                    (4, "leap-frog + constraints"), (8, "fused update + sorted coordinates")):
    print("%-36s %.2f us" % (name, e.debug_time_stage(stage, 200) * 1e3), flush=True)
