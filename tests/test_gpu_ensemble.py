"""Config 4 of the benchmark contract on the GPU: the SMD workload (pull_short.mdp settings, constant-force COM pull)
run through bench.py with the `nccl` (RCCL) process group at world size 1, so that the all-gather of the pull work and
the Jarzynski average take the device path they take on eight GPUs (/root/reference/Snakefile:753-802 is the per-pose
fan-out the ensemble replaces). The gathered work and the free-energy estimate are checked against their CPU
evaluation."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_smd_workload_through_rccl_all_gather():
    env = dict(os.environ, MDX_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29600 + os.getpid() % 300), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "smd", "--box", "8.0", "--prep-steps", "300",
           "--steps", "400", "--warmup", "20", "--prime-steps", "20", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["parallelism"] == "replicas x1" and out["scaling"] == "weak"
    smd = out["smd"]
    assert smd["replicas"] == 1 and smd["samples"] >= 8
    # constant force: W = -k (r - r0), accumulated on the device, gathered through RCCL
    w = smd["work_final_all"]
    assert len(w) == 1 and w[0] == smd["work_final_rank0"]
    assert abs(w[0] + smd["k"] * (smd["final_distance"] - smd["initial_distance"])) < 1e-6 * smd["k"]
    # one replica: the Jarzynski estimate -kT ln <exp(-W/kT)> is the work itself
    assert abs(smd["jarzynski_dF_final"] - w[0]) < 1e-9 * max(1.0, abs(w[0]))
    # attraction (k > 0 pulls the ligand towards the site): the distance shrinks, the work done on the system is positive
    assert smd["final_distance"] < smd["initial_distance"]
    assert out["value"] > 0 and out["roofline"]["launches"] > 0


def test_bench_refuses_more_gpus_than_present():
    import torch
    n = torch.cuda.device_count()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--workload", "smd", "--steps", "10"],
                       capture_output=True, text=True, timeout=120, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    assert p.returncode != 0 and "needs %d GPUs" % (n + 1) in p.stderr
