#!/usr/bin/env python3
"""Real timeline of MD steps from in-kernel time stamps (library built with -DMDX_STAMPS: make ../libmdx_ts.so).
No tracing tool sits between the streams, and the host is several steps ahead of the GPU when the stamped
step runs. Usage (GPU box): MDX_LIB=molecular-dynamics_amd/libmdx_ts.so python tools/step_stamps.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
NAMES = {0: "prepare", 1: "nonbonded", 2: "nonbonded (energy)", 3: "spread", 4: "fft x", 5: "fft y (or y+z fused)", 6: "fft z+solve", 7: "fft y inv",
         8: "fft x inv", 9: "gather", 10: "bonded", 11: "leapfrog / fused update", 12: "settle", 13: "lincs", 15: "tcouple", 17: "build list", 18: "prune list"}
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
DUAL = bool(os.environ.get("MDX_DUAL"))       # the bench's dual pair list: outer list 100 steps, pruned every 10 (beside the energy steps)
CYC = 100 if DUAL else 20
e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=CYC, nstprune=10 if DUAL else 0, nstcalcenergy=10, tc_groups=tc, ref_t=[300, 300], tau_t=[0.1, 0.1], nsttcouple=10,
               pcoupl=1, tau_p=2.0, ref_p=1.0, compressibility=4.5e-5, nstpcouple=10, dispcorr=2)
for kv in filter(None, os.environ.get("MDX_DEBUG", "").split(",")):
    e.debug_set(int(kv.split("=")[0]), int(kv.split("=")[1], 0))
e.step(2000); e.synchronize()
for kv in filter(None, os.environ.get("MDX_DEBUG_LATE", "").split(",")):      # switches that break the physics: only for the stamped steps
    e.debug_set(int(kv.split("=")[0]), int(kv.split("=")[1], 0))
def stamped(nlead, nstamp, ntrail):
    """nlead steps, reset, nstamp stamped steps, ntrail more steps, all enqueued back to back"""
    e.synchronize()
    if nlead > 0:
        # two steps in one call: the stamps are plain stores, so they show the second step, whose side streams were
        # released by its predecessor in the same call (as in a long run)
        e.step(nlead - 1); e.debug_stamps(reset=True); e.step(nstamp + 1)
    else:
        e.debug_stamps(reset=True); e.step(nstamp)
    t = e.debug_stamps()
    return t
# align to the step counter: plain steps are those with step % 10 not in (0, 9) and no list rebuild
st = e.get_step()
e.step((CYC - st % CYC) % CYC); e.synchronize()      # now step % CYC == 0: next step is search + energy
for lead, label in ((3, "plain step (step %% 20 == 3)"), (10, "energy step (step %% 20 == 10)"), (0, "search + energy step (step %% 20 == 0)")):
    e.synchronize(); st = e.get_step(); e.step((CYC - st % CYC) % CYC); e.synchronize()
    t = stamped(lead, 1, 0)
    t0 = t[0][0] if t[1][0] > 0 else min(v for v in t[0] if v < 2**62)
    # a plain step that follows a fused update in the same call has no prepare launch: the prepare stamp is then that of
    # an earlier step and the timeline is counted from the start of the step's first kernel instead
    nbk = 2 if (t[1][2] > t[1][1]) else 1
    if t[1][0] > 0 and t[1][0] < t[0][nbk] - 3000:
        t0 = min(t[0][k] for k in range(1, 24) if t[1][k] > t[0][nbk] - 3000 and t[0][k] < 2**62 and t[0][k] > t[1][0])
    print("---- %s" % label)
    # (kernels that only the first of the two steps launched keep its stamps: they end before this step's prepare starts)
    rows = [(t[0][k], t[1][k], NAMES.get(k, str(k))) for k in range(24) if t[1][k] > 0 and (t[1][k] >= t0 or lead == 0 or k == 3) and t[0][k] >= t0 - 40000]
    for a, b, n in sorted(rows):
        print("%8.1f %8.1f  %6.1f  %s" % ((a - t0) / 100.0, (b - t0) / 100.0, (b - a) / 100.0, n))
# average step period from stamps over 40 steps: prepare start of first vs last is not kept, so time it from the host
import time
if os.environ.get("MDX_DEBUG_LATE"): sys.exit(0)
e.synchronize(); t0 = time.perf_counter(); e.step(2000); e.synchronize(); print("2000 steps: %.1f us/step" % ((time.perf_counter() - t0) / 2000 * 1e6))
