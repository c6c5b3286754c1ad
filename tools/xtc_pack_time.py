"""Host-side timing of the xtc bit packing (coordinates rounded beforehand, as the engine's capture delivers them) on a
water box of the bench size: ms per frame against threads per frame, with the time of the parallel segments, of the merge
and of the slowest segment."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mdx  # noqa: E402,F401
from mdx import synth  # noqa: E402
from mdx.io import traj  # noqa: E402
s = synth.water_box(d=12.5)
x = s.x.astype(np.float32); box = s.box.astype(np.float32)
lf = x * np.float32(1000.0)
ix = np.where(lf >= 0, lf + np.float32(0.5), lf - np.float32(0.5)).astype(np.int32)
d = np.abs(np.diff(ix.astype(np.int64), axis=0)).sum(1)
stats = np.array(list(ix.min(0)) + list(ix.max(0)) + [int(d.min()), 0], dtype=np.int32)
L = traj.load_library()
L.trj_xtc_last_phases.argtypes = [ctypes.POINTER(ctypes.c_double)]
ph = (ctypes.c_double * 20)()
for T in (1, 2, 3, 4, 6, 8, 12, 16):
    w = traj.XtcWriter(os.devnull, s.natoms, threads=T)
    w.encode_quantized(0, 0.0, box, ix, stats)
    ts, pp = [], []
    for k in range(40):
        t0 = time.perf_counter(); w.encode_quantized(k, 0.0, box, ix, stats); ts.append(time.perf_counter() - t0)
        L.trj_xtc_last_phases(ph); pp.append(list(ph))
    ts = np.array(ts) * 1e3; pp = np.median(np.array(pp), axis=0)
    print("T=%2d: median %.2f ms min %.2f | segments %.2f merge %.2f slowest segment %.2f" % (T, np.median(ts), ts.min(), pp[0], pp[1], pp[3]), " per segment:", " ".join("%.2f" % v for v in pp[4:4 + T]), flush=True)
