import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
e = eng.Engine(s, rc=1.0, dt=0.0005, nstlist=20, nstcalcenergy=0, tc_groups=["Protein", "Non-Protein"], ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2)
e.step(200); e.synchronize()
t = e.debug_wave_times()
xcc = t[:, 7] & 0xF
blk = np.arange(len(xcc)) // 4
print("waves", len(xcc), "blocks", blk.max() + 1)
print("xcc == block % 8 for", int((xcc == blk % 8).sum()), "of", len(xcc))
for b in range(0, 24): print(b, xcc[4 * b: 4 * b + 4])
