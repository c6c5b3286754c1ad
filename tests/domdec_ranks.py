#!/usr/bin/env python3
"""Helper of test_gpu_domdec.py (also runnable by hand on a GPU box): `world` domains on cuda:0, one thread each in
this one process (domdec.ThreadTransport; the torch.distributed transport is exercised on CPU by
test_domdec_partition.py). Runs the decomposed water box and the single-handle run of the same box and prints one
JSON line with
 - force parity: one step from rest (v = 0), so the new velocities are dt * f / m plus the constraint correction:
   relative RMS difference of the velocities between the two runs;
 - trajectory agreement after `steps` steps (pair-list lifetime 10: the partition is renewed on the way).
python tests/domdec_ranks.py [world] [steps] [box d]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def decomposed(world, nst, rest, d, migration=True, mesh="slabs"):
    import mdx
    from mdx import synth, domdec
    hub = domdec.ThreadTransport.Hub(world, exclusive=bool(os.environ.get("MDX_DD_EXCLUSIVE")))

    def rank_main(rank, transport):
        s = synth.water_box(d=d)
        if rest:
            s.v = np.zeros_like(s.v)
        dd = domdec.DecomposedWater(s, device=0, rc=1.0, dt=0.002, nstlist=10, ref_t=None, rlist=1.1, transport=transport,
                                    migration=migration, mesh=mesh)
        info = dict(dims=dd.dims, home=dd.n_home, halo=dd.n_halo, cap=dd.cap, mesh=dd.mesh)
        dd.step(nst); dd.synchronize()
        info["list_tiles"] = dd.engine.list_stats()["cluster_pairs"]
        x, v = dd.gather_state()
        dd.close()
        return dict(x=x, v=v, info=info)
    return hub.run(rank_main)[0]


def coupled(world, nst, d):
    """Thermostat, centre-of-mass removal and energies across the domains against the single handle (same seed)."""
    import mdx
    from mdx import synth, domdec, engine as eng
    hub = domdec.ThreadTransport.Hub(world)
    kw = dict(rc=1.0, dt=0.002, nstlist=10, rlist=1.1)

    def rank_main(rank, transport):
        dd = domdec.DecomposedWater(synth.water_box(d=d), device=0, ref_t=300.0, tau_t=0.1, nsttcouple=10, nstcalcenergy=10,
                                    nstcomm=10, seed=7, transport=transport, **kw)
        dd.step(nst); dd.synchronize()
        e, vir, pres = dd.energies()
        x, v = dd.gather_state()
        dd.close()
        return dict(e=e, vir=vir, pres=pres, v=v, mesh=dd.mesh)
    o = hub.run(rank_main)[0]
    s = synth.water_box(d=d)
    e = eng.Engine(s, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1], nsttcouple=10, nstcalcenergy=10, nstcomm=10, seed=7, **kw)
    e.step(nst); e.synchronize()
    er = e.energies(); virr, presr = e.virial()
    _, vr, _ = e.get_state(); e.close()
    terms = ("LJ", "COUL_SR", "COUL_RECIP", "COUL_EXCL", "KIN", "TEMP", "PRES", "POT")
    return dict(dd={k: o["e"][k] for k in terms}, single={k: er[k] for k in terms},
                vir_max_diff=float(np.abs(o["vir"] - virr).max()), vir_scale=float(np.abs(virr).max()),
                pres_max_diff=float(np.abs(o["pres"] - presr).max()),
                dv_rel_rms=float(np.sqrt(((o["v"] - vr) ** 2).mean() / (vr ** 2).mean())),
                com_v=[float(c) for c in (s.mass[:, None] * o["v"]).sum(0) / s.mass.sum()], mesh=o["mesh"])


def coupled_npt(world, nst, d, kind="c-rescale"):
    """C-rescale pressure coupling across the domains against the single handle: same seed, so the same stochastic
    scaling factors as long as the summed pressure agrees; the cell after `nst` steps (three coupling steps) and the
    positions are compared."""
    import mdx
    from mdx import synth, domdec, engine as eng
    hub = domdec.ThreadTransport.Hub(world)
    kw = dict(rc=1.0, dt=0.002, nstlist=10, rlist=1.1)
    pk = dict(tau_p=1.0, ref_p=1.0, compressibility=4.5e-5, nstpcouple=10, dispcorr=2)     # DispCorr = EnerPres as in production.mdp

    def rank_main(rank, transport):
        dd = domdec.DecomposedWater(synth.water_box(d=d), device=0, ref_t=300.0, tau_t=0.1, nsttcouple=10, nstcalcenergy=10,
                                    seed=11, transport=transport, pcoupl=kind, **pk, **kw)
        dd.step(nst); dd.synchronize()
        e, vir, pres = dd.energies()
        x, v = dd.gather_state()
        box = dd.box_np.copy()
        dd.close()
        return dict(e=e, pres=pres, x=x, box=box)
    o = hub.run(rank_main)[0]
    s = synth.water_box(d=d)
    e = eng.Engine(s, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1], nsttcouple=10, nstcalcenergy=10, nstcomm=0, seed=11,
                   pcoupl=1 if kind == "c-rescale" else 2, **pk, **kw)
    e.step(nst); e.synchronize()
    er = e.energies(); _, presr = e.virial()
    xr, _, boxr = e.get_state(); e.close()
    boxr = np.asarray(boxr, dtype=np.float64).reshape(3, 3)
    dx = o["x"] - xr
    dx -= np.round(dx @ np.linalg.inv(boxr)) @ boxr
    return dict(box_rel_diff=float(np.abs(o["box"] - boxr).max() / np.abs(boxr).max()),
                box_change=float(abs(boxr[0, 0] / s.box[0][0] - 1.0)), dx_rms=float(np.sqrt((dx ** 2).mean())),
                pot_dd=o["e"]["POT"], pot_single=er["POT"], pres_max_diff=float(np.abs(o["pres"] - presr).max()),
                dispcorr_dd=o["e"]["DISPCORR"], dispcorr_single=er["DISPCORR"])


def main(world=2, steps=12, d=7.0):
    import faulthandler; faulthandler.dump_traceback_later(int(os.environ.get("MDX_DD_LIMIT", "150")), exit=True)
    import mdx
    from mdx import synth, engine as eng
    res = {"world": world, "steps": steps}
    # ("traj_gather": the partition renewed by gathering every molecule on every rank instead of by migration, and the
    # real-space charge grids summed on every rank instead of the mesh transform shared by planes)
    for label, nst, rest in (("rest", 1, True), ("traj", steps, False), ("traj_gather", steps, False)):
        o = decomposed(world, nst, rest, d, migration=label != "traj_gather", mesh="replicated" if label == "traj_gather" else "slabs")
        s = synth.water_box(d=d)
        if rest:
            s.v = np.zeros_like(s.v)
        e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=10, rlist=1.1, nstcalcenergy=0, nstcomm=0)
        e.step(nst); e.synchronize()
        xr, vr, _ = e.get_state(); e.close()
        dx = o["x"] - xr
        dx -= np.round(dx @ np.linalg.inv(s.box)) @ s.box
        res[label] = dict(dx_max=float(np.abs(dx).max()), dx_rms=float(np.sqrt((dx ** 2).mean())),
                          dv_rel_rms=float(np.sqrt(((o["v"] - vr) ** 2).mean() / (vr ** 2).mean())), partition=o["info"])
    res["coupled"] = coupled(world, 21, d)
    res["npt"] = coupled_npt(world, 21, d)
    res["npt_pr"] = coupled_npt(world, 21, d, "parrinello-rahman")
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if len(a) > 0 else 2, int(a[1]) if len(a) > 1 else 12, float(a[2]) if len(a) > 2 else 7.0)
