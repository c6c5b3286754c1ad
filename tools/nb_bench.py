#!/usr/bin/env python3
"""Nonbonded launch-shape A/B: alone-time of the force kernel per variant on the S2 list and a
one-step consistency check of every variant against the default one.
Variant = 10 * (waves/SIMD of the energy kernel) + force-kernel digit (2-4 scalar body at that
many waves/SIMD, 7 / 8 packed-math body at 3 / 4 waves/SIMD)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng

variants = [int(a) for a in sys.argv[1:]] or [33, 34, 37, 38]
s = synth.pseudo_complex(d=6.0, protein_radius=1.0, n_active=60, ligand_offset=1.2)
ref = None
for v in variants:
    e = eng.Engine(s, rc=1.0, nstcalcenergy=0, nstlist=10)
    e.debug_set(1, v)
    e.step(3)
    x, vel, _ = e.get_state()
    e.close()
    if ref is None:
        ref = (x, vel)
    print("variant %d: 3 steps, max |dx| vs first variant %.2e nm, max |dv| %.2e nm/ps" % (v, np.abs(x - ref[0]).max(), np.abs(vel - ref[1]).max()))

s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstlist=20, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2)
e.step(1500); e.synchronize()
for rep in range(2):
    for v in variants:
        e.debug_set(1, v)
        print("variant %d: force kernel alone %.1f us" % (v, e.debug_time_nonbonded(False, 40) * 1e3))
