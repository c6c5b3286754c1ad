import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=6.0, protein_radius=1.0, n_active=60, ligand_offset=1.2)
s.x = s.x.astype(np.float32).astype(np.float64)
v0 = s.v.astype(np.float32).astype(np.float64)
res = []
for bits in (0, 0x8000000):
    e = eng.Engine(s, rc=1.0, nstcalcenergy=0, nstcomm=0, dt=0.002)
    e.debug_set(2, bits)
    e.set_state(s.x, v0, s.box); e.step(1)
    x, v, _ = e.get_state(); res.append((x.astype(np.float64), v.astype(np.float64))); e.close()
print("max |dx| %.3e  max |dv| %.3e (packed against plain force kernel, one step)" % (np.abs(res[0][0]-res[1][0]).max(), np.abs(res[0][1]-res[1][1]).max()))
