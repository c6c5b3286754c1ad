#!/usr/bin/env python3
"""Where the cost of a trajectory frame every 10 steps goes on the bench workload. Step period of
  a) steps in calls of 10, no capture;
  b) + capture into the ring (rounding kernel, device-to-host copy), nobody waits for the frames;
  c) + a thread that waits for every frame's copy (mdx_frame_wait_xtc) and does nothing with it;
  d) + that thread packs the frame (4 packing threads) and writes it: the bench's pipeline.
b - a is the GPU side of a capture, c - b what a host thread waiting on the copy's event costs the stepping thread,
d - c the packing threads. Usage (GPU box): python tools/frame_cost.py"""
import os, sys, time, threading, queue, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
from mdx.io import traj
from mdx.output import FramePipeline

s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=100, nstprune=10, tc_groups=tc, ref_t=[300, 300], tau_t=[0.1, 0.1], nstcalcenergy=10, nsttcouple=10,
               pcoupl=2, tau_p=2.0, ref_p=1.0, compressibility=4.5e-5, nstpcouple=10, dispcorr=2)
for kv in filter(None, os.environ.get("MDX_DEBUG", "").split(",")):
    e.debug_set(int(kv.split("=")[0]), int(kv.split("=")[1], 0))
e.step(3000); e.synchronize()
NSLOT = 8
e.frame_ring(NSLOT)
NFR = 400


def loop(capture):
    e.synchronize()
    t0 = time.perf_counter()
    for i in range(NFR):
        capture(i)
        e.step(10)
    e.synchronize()
    return (time.perf_counter() - t0) / (NFR * 10) * 1e6


if os.environ.get("FRAME_COST_TRACE"):       # under rocprofv3 --kernel-trace --memory-copy-trace: a short run of b) only
    NFR = 30
    loop(lambda i: e.capture_frame_xtc(i % NSLOT, 1000.0, want_v=False))
    e.close(); sys.exit(0)
for rep in range(2):
    a = loop(lambda i: None)
    b = loop(lambda i: e.capture_frame_xtc(i % NSLOT, 1000.0, want_v=False))
    # c: a waiting thread, slots handed back through a queue (back-pressure as in the pipeline)
    free = queue.Queue(); todo = queue.Queue()
    for k in range(NSLOT):
        free.put(k)

    def waiter():
        while True:
            k = todo.get()
            if k is None:
                return
            e.frame_wait_xtc(k)
            free.put(k)
    th = threading.Thread(target=waiter, daemon=True); th.start()

    def cap_c(i):
        k = free.get(); e.capture_frame_xtc(k, 1000.0, want_v=False); todo.put(k)
    c = loop(cap_c)
    todo.put(None); th.join()
    path = os.path.join(tempfile.gettempdir(), "frame_cost_%d.xtc" % os.getpid())
    xtc = traj.XtcWriter(path, s.natoms, precision=1000.0, threads=4)
    pipe = FramePipeline(e, NSLOT, xtc, None, None, 0.002, 0.0, nthreads=1)
    d = loop(lambda i: pipe.capture(True, False, False))
    pipe.finish(); xtc.close(); os.remove(path)
    print("pass %d: a) no capture %.2f  b) capture, nobody waits %.2f  c) + waiting thread %.2f  d) + packing and writing %.2f us/step" % (rep, a, b, c, d), flush=True)
e.close()
