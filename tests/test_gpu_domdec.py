"""Spatial domain decomposition on the GPU: two ranks (sharing the one device, gloo between them) against the
single-handle run of the same water box. The reference has no decomposed runs (gmx mdrun -ntmpi 1,
/root/reference/Snakefile:232-233): the single handle, itself checked against the oracle, is the yardstick."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_domains_match_single_handle():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "domdec_ranks.py"), "2", "12", "7.0"],
                       capture_output=True, text=True, timeout=170)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    part = res["partition"]["traj"]
    assert part["halo"] > 0 and part["home"] > 0
    # forces through the whole decomposed path (halo exchange, summed charge grid): one step from rest, float32
    assert res["rest"]["dv_rel_rms"] < 1e-5, res
    # 12 steps with one renewal of the partition: rounding-level differences, amplified by the dynamics
    assert res["traj"]["dx_rms"] < 5e-6 and res["traj"]["dx_max"] < 2e-4, res
