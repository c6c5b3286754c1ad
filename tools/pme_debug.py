import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
from oracle import mdref as R
s = synth.water_box(d=3.4); s.x = s.x.astype(np.float32).astype(np.float64)
e = eng.Engine(s, rc=1.0, nstcalcenergy=1)
f, en = e.compute(eng.F_PME)
Ep, Fp, vp = R.pme_recip(s.x, s.charge, s.box, e.beta, e.grid)
print("grid", e.grid, "E", en["COUL_RECIP"], Ep)
err = np.linalg.norm(f - Fp, axis=1)
idx = np.argsort(-err)[:15]
inv = np.linalg.inv(s.box)
for i in idx:
    sfrac = (s.x[i] @ inv) % 1.0
    print(i, err[i], np.linalg.norm(Fp[i]), sfrac * np.array(e.grid))
print("median err", np.median(err), "mean |F|", np.linalg.norm(Fp, axis=1).mean())
