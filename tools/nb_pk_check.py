import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=6.0, protein_radius=1.0, n_active=60, ligand_offset=1.2)
s.x = s.x.astype(np.float32).astype(np.float64)
v0 = s.v.astype(np.float32).astype(np.float64)
res = []
bits_list = [0] + ([int(a, 0) for a in sys.argv[1:]] or [0x20000000])
for bits in bits_list:
    e = eng.Engine(s, rc=1.0, nstcalcenergy=0, nstcomm=0, dt=0.002)
    e.debug_set(2, bits)
    e.set_state(s.x, v0, s.box); e.step(1)
    x, v, _ = e.get_state(); res.append((x.astype(np.float64), v.astype(np.float64))); e.close()
for b, r in zip(bits_list[1:], res[1:]):
    print("bits 0x%x: max |dx| %.3e  max |dv| %.3e against the default force kernel, one step" % (b, np.abs(res[0][0]-r[0]).max(), np.abs(res[0][1]-r[1]).max()))
