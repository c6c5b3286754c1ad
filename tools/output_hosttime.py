#!/usr/bin/env python3
"""Where the stepping thread spends host time when frames are captured every 10 steps (S2 system)."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
if os.environ.get("WITH_TORCH"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
from mdx import synth, engine as eng
from mdx.io import traj
from mdx.output import FramePipeline
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstlist=20, nstcalcenergy=10, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=10)
e.step(1000); e.synchronize()
for mode, bits in (("none", 0), ("capture", 0), ("capture", 32768), ("capture", 65536), ("full", 0), ("full", 65536)):
    e.debug_set(2, bits)
    path = os.path.join(tempfile.gettempdir(), "ht.xtc")
    xtc = traj.XtcWriter(path, s.natoms, precision=1000.0)
    e.frame_ring(8)
    pipe = FramePipeline(e, 8, xtc, None, None, 0.002, 0.0, nthreads=3)
    T = dict(acq=0.0, cap=0.0, sub=0.0, step=0.0)
    e.synchronize(); t0 = time.perf_counter()
    done = 0
    while done < 2000:
        if mode != "none":
            a = time.perf_counter()
            slot = pipe.acquire_slot() if mode == "full" else (done // 10) % 8
            b = time.perf_counter(); T["acq"] += b - a
            e.capture_frame(slot)
            c = time.perf_counter(); T["cap"] += c - b
            if mode == "full": pipe.submit(slot, True, False, False)
            T["sub"] += time.perf_counter() - c
        a = time.perf_counter()
        e.step(10); done += 10
        T["step"] += time.perf_counter() - a
    th = time.perf_counter() - t0
    e.synchronize(); el = time.perf_counter() - t0
    pipe.finish(); xtc.close()
    print("%-8s bits %6d wall %.1f us/step (host loop %.1f us/step)  per 10 steps: acquire %.0f us  capture %.0f us  submit %.0f us  step call %.0f us" % (
        mode, bits, el / 2000 * 1e6, th / 2000 * 1e6, T["acq"] / 200 * 1e6, T["cap"] / 200 * 1e6, T["sub"] / 200 * 1e6, T["step"] / 200 * 1e6), flush=True)
