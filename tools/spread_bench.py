#!/usr/bin/env python3
"""Spread-kernel A/B: alone-time of each variant on the S2 system and PME force parity on a small box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
from oracle import mdref as R

s = synth.pseudo_complex(d=6.0, protein_radius=1.0, n_active=60, ligand_offset=1.2)
s.x = s.x.astype(np.float32).astype(np.float64)
e = eng.Engine(s, rc=1.0, nstcalcenergy=1)
Ep, Fp, vp = R.pme_recip(s.x, s.charge, s.box, e.beta, e.grid)
for bits in (0,):
    e.debug_set(2, bits)
    f, en = e.compute(eng.F_PME)
    print("dbg %4d  PME force relrms %.3e  E rel %.3e" % (bits, np.sqrt(((f - Fp) ** 2).sum() / (Fp ** 2).sum()), abs(en["COUL_RECIP"] - Ep) / abs(Ep)))
e.close()

s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2)
e.step(1500); e.synchronize()
for bits, label in ((0, "LDS tile per column"),):
    e.debug_set(2, bits)
    print("%-32s spread %.1f us   chain %.1f us" % (label, e.debug_time_stage(6, 30) * 1e3, e.debug_time_stage(2, 30) * 1e3))
