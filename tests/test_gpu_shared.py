"""Two `mdrun`-like processes on one GPU, as Snakemake starts them when several poses run at once
(/root/reference/Snakefile:520-802 fans the per-pose chains out over the `{pose}` wildcard): both have to make
progress at normal speed. (With side streams of a higher priority than the main stream, as an earlier version of the
engine created them, the hardware queues of the two processes keep pre-empting each other's pair kernel and a step
takes tens of milliseconds to seconds; all streams of a handle now have the default priority.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_two_processes_share_one_gpu_at_normal_speed(tmp_path):
    helper = os.path.join(ROOT, "tests", "shared_gpu_ranks.py")
    env = dict(os.environ, LIMIT="100")
    env.pop("MDX_STREAMS", None)
    procs = [subprocess.Popen([sys.executable, helper, str(r), str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in (0, 1)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a process sharing the GPU did not finish")
        assert p.returncode == 0, o[-1000:] + e[-3000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
    res, new = outs
    # both kept their three streams (the PME and bonded chains overlap the pair kernel as on a GPU of their own)
    assert set(new["modes"]) == {3} and set(res["modes"]) == {3}
    assert res["n_shared"] > 20 and new["chunks"] == 60
    # nobody crawled: a 10-step chunk of this 17 k-atom box takes a few milliseconds; the pathological case is >0.5 s
    assert new["worst_chunk_s"] < 0.25 and res["worst_chunk_s"] < 0.25, (res["worst_chunk_s"], new["worst_chunk_s"])


def test_torch_reaches_the_gpu_after_the_engine_was_loaded_first():
    """One HIP runtime per process: the engine's loader brings in the runtime a PyTorch-ROCm installation ships before
    libmdx.so, so that a program that creates an engine first and touches torch.cuda afterwards does not end up with
    two runtimes (torch then reports "No HIP GPUs are available")."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hip_runtime_order.py")], capture_output=True, text=True, timeout=150)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
    assert "torch on the GPU after the engine: ok" in p.stdout, p.stdout[-2000:]
