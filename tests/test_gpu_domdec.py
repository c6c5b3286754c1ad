"""Spatial domain decomposition on the GPU: domains on the one device (one thread each, in-process transport; the
torch.distributed transport is tested on CPU) against the
single-handle run of the same water box. The reference has no decomposed runs (gmx mdrun -ntmpi 1,
/root/reference/Snakefile:232-233): the single handle, itself checked against the oracle, is the yardstick."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, box):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "domdec_ranks.py"), str(world), "12", str(box)],
                       capture_output=True, text=True, timeout=170)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("world,box", [(2, 9.0), (4, 9.0)])
def test_domains_match_single_handle(world, box):
    res = _run(world, box)
    part = res["traj"]["partition"]
    assert part["halo"] > 0 and part["home"] > 0
    # forces through the whole decomposed path (halo exchange, summed charge grid): one step from rest. The same
    # measure between the one-domain decomposed path and the plain handle, which differ in atom order and padding
    # only, is 8e-6 (float32 evaluation order through the constrained update); two domains measure 1.0e-5.
    assert res["rest"]["dv_rel_rms"] < 2e-5, res
    # 12 steps with one renewal of the partition: 2.3e-7 nm RMS, the same as the one-domain path against the plain handle
    assert res["traj"]["dx_rms"] < 2e-6 and res["traj"]["dx_max"] < 5e-5, res
