"""One rank of the decomposed path inside an RCCL process group of size 1, with every reduction issued for real
(MDX_DD_FORCE_COLLECTIVES=1) and ordered on the handle's streams instead of by host synchronisation; compared with the
plain single handle. Run by tests/test_gpu_domdec.py as a child process."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(d=7.0, nst=21):
    import torch
    import torch.distributed as dist
    os.environ["MDX_DD_FORCE_COLLECTIVES"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    first = torch.ones(1, device="cuda"); dist.all_reduce(first); torch.cuda.synchronize()      # communicator complete ahead of the handle
    import mdx  # noqa: F401
    from mdx import synth, domdec, engine as eng
    kw = dict(rc=1.0, dt=0.002, nstlist=10, rlist=1.1)
    pk = dict(tau_p=1.0, ref_p=1.0, compressibility=4.5e-5, nstpcouple=10, dispcorr=2)
    dd = domdec.DecomposedWater(synth.water_box(d=d), device=0, ref_t=300.0, tau_t=0.1, nsttcouple=10, nstcalcenergy=10, seed=11,
                                pcoupl="c-rescale", nstcomm=10, **pk, **kw)
    assert dd.ordered and dd.world == 1
    dd.step(nst); dd.synchronize()
    e, vir, pres = dd.energies()
    x, v = dd.gather_state()
    box = dd.box_np.copy()
    dd.close()
    s = synth.water_box(d=d)
    ref = eng.Engine(s, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1], nsttcouple=10, nstcalcenergy=10, nstcomm=10, seed=11, pcoupl=1, **pk, **kw)
    ref.step(nst); ref.synchronize()
    er = ref.energies(); _, presr = ref.virial()
    xr, vr, boxr = ref.get_state(); ref.close()
    boxr = np.asarray(boxr, dtype=np.float64).reshape(3, 3)
    dx = x - xr
    dx -= np.round(dx @ np.linalg.inv(boxr)) @ boxr
    out = dict(box_rel_diff=float(np.abs(box - boxr).max() / np.abs(boxr).max()), box_change=float(abs(boxr[0, 0] / s.box[0][0] - 1.0)),
               dx_rms=float(np.sqrt((dx ** 2).mean())), dv_rel_rms=float(np.sqrt(((v - vr) ** 2).mean() / (vr ** 2).mean())),
               pot_dd=e["POT"], pot_single=er["POT"], pres_max_diff=float(np.abs(pres - presr).max()),
               com_v=[float(c) for c in (s.mass[:, None] * v).sum(0) / s.mass.sum()])
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
