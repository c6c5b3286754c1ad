#!/usr/bin/env python3
"""Helper of test_gpu_domdec.py (also runnable by hand on a GPU box): `world` ranks share cuda:0 and talk through
gloo; runs the decomposed water box and the single-handle run of the same box and prints one JSON line with
 - force parity: one step from rest (v = 0), so the new velocities are dt * f / m plus the constraint correction:
   relative RMS difference of the velocities between the two runs;
 - trajectory agreement after `steps` steps (pair-list lifetime 10: the partition is renewed on the way).
python tests/domdec_ranks.py [world] [steps] [box d]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch.distributed as dist, torch.multiprocessing as mp


def run(rank, world, steps, d, out, port):
    import faulthandler; faulthandler.dump_traceback_later(170, exit=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mdx
    from mdx import synth, domdec
    res = {}
    for label, nst, rest in (("rest", 1, True), ("traj", steps, False)):
        s = synth.water_box(d=d)
        if rest:
            s.v = np.zeros_like(s.v)
        dd = domdec.DecomposedWater(s, device=0, rc=1.0, dt=0.002, nstlist=10, ref_t=None, rlist=1.1, shared_device=True)
        info = dict(dims=dd.dims, home=dd.n_home, halo=dd.n_halo, cap=dd.cap)
        dd.step(nst); dd.synchronize()
        x, v = dd.gather_state()
        if rank == 0:
            np.save(out + "_" + label + ".npy", np.stack([x, v]))
            res[label] = info
        dd.close()
    if rank == 0:
        json.dump(res, open(out + "_info.json", "w"))
    dist.destroy_process_group()


def main(world=2, steps=12, d=7.0):
    out = "/tmp/mdx_dd_%d" % os.getpid()
    mp.spawn(run, args=(world, steps, d, out, 29500 + os.getpid() % 400), nprocs=world, join=True)
    import mdx
    from mdx import synth, engine as eng
    info = json.load(open(out + "_info.json"))
    res = {"world": world, "steps": steps, "partition": info}
    for label, nst, rest in (("rest", 1, True), ("traj", steps, False)):
        s = synth.water_box(d=d)
        if rest:
            s.v = np.zeros_like(s.v)
        e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=10, rlist=1.1, nstcalcenergy=0, nstcomm=0)
        e.step(nst); e.synchronize()
        xr, vr, _ = e.get_state(); e.close()
        xd, vd = np.load(out + "_" + label + ".npy")
        dx = xd - xr
        dx -= np.round(dx @ np.linalg.inv(s.box)) @ s.box
        res[label] = dict(dx_max=float(np.abs(dx).max()), dx_rms=float(np.sqrt((dx ** 2).mean())),
                          dv_rel_rms=float(np.sqrt(((vd - vr) ** 2).mean() / (vr ** 2).mean())))
    for f in ("_rest.npy", "_traj.npy", "_info.json"):
        os.remove(out + f)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if len(a) > 0 else 2, int(a[1]) if len(a) > 1 else 12, float(a[2]) if len(a) > 2 else 7.0)
