#!/usr/bin/env python3
"""Helper of test_gpu_shared.py: one of two engine processes on the same GPU. The resident (role 0) starts alone
and keeps stepping; the newcomer (role 1) starts once the resident has run for a while, runs its own
steps and leaves; the resident then carries on alone. Each prints one JSON line: wall time of every 10-step chunk
and the stream mode the handle reported after each chunk.
python tests/shared_gpu_ranks.py ROLE FLAGDIR"""
import json, os, sys, time, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
faulthandler.dump_traceback_later(int(os.environ.get("LIMIT", "100")), exit=True)
role, flagdir = int(sys.argv[1]), sys.argv[2]


def flag(name):
    return os.path.join(flagdir, name)


def wait_for(name, limit=60.0):
    t0 = time.time()
    while not os.path.exists(flag(name)):
        if time.time() - t0 > limit:
            raise SystemExit("timed out waiting for " + name)
        time.sleep(0.005)


import mdx  # noqa: E402,F401
from mdx import synth, engine as eng  # noqa: E402
s = synth.water_box(d=6.0)
if role == 1:
    wait_for("resident_running")
e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=10, nstcalcenergy=0, nstcomm=0)
chunks, modes = [], []


def run(nchunks):
    for _ in range(nchunks):
        t0 = time.time()
        e.step(10); e.synchronize()
        chunks.append(time.time() - t0)
        modes.append(e.list_stats()["streams"])


if role == 0:
    run(20)
    open(flag("resident_running"), "w").close()
    while not os.path.exists(flag("newcomer_done")):
        run(5)
        if len(chunks) > 4000:
            break
    n_shared = len(chunks)
    run(40)
else:
    run(60)
    n_shared = len(chunks)
    x, v, _ = e.get_state()
    e.close()
    open(flag("newcomer_done"), "w").close()
print(json.dumps(dict(role=role, chunks=len(chunks), n_shared=n_shared, worst_chunk_s=max(chunks), total_s=sum(chunks), modes=modes)), flush=True)
