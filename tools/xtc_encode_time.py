"""Host-side timing of the xtc encoder on a water box of the bench size: ms per frame against threads per frame."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mdx  # noqa: E402,F401
from mdx import synth  # noqa: E402
from mdx.io import traj  # noqa: E402

s = synth.water_box(d=12.5)
x = s.x.astype(np.float32)
box = s.box.astype(np.float32)
for T in (1, 2, 4, 6, 8, 12, 16):
    w = traj.XtcWriter(os.devnull, s.natoms, precision=1000.0, threads=T)
    w.encode(0, 0.0, box, x)
    best, tot = 1e9, 0.0
    for k in range(20):
        t0 = time.perf_counter()
        b = w.encode(k, 0.0, box, x)
        dt = time.perf_counter() - t0
        best = min(best, dt); tot += dt
    print("%2d threads: %.2f ms/frame mean, %.2f best, %d bytes" % (T, tot / 20 * 1e3, best * 1e3, len(b)), flush=True)
    w.close()
