#!/bin/bash
# Everything the committed profiles under profiles/r03 are made from (GPU box, repository root): the GPU tests, the smoke
# entry, the default bench line (with its command-line leg), the driver's 20-step invocation, the kernel trace statistics
# of the bench command, the in-kernel step timeline on the dual pair list, and the counter passes of the nonbonded
# kernel. Each GPU step has its own time limit; the script stops at the first failure.
set -e
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O; rm -rf $O/prof
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_gpu.log 2>&1 || { tail -20 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 200 python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_20_steps_driver_invocation.json 2> $O/bench_20.err || { tail -5 $O/bench_20.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $R/bench.py --steps 2000 --warmup 300 --no-cpu-baseline --no-cli > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
cd $R
MDX_DUAL=1 MDX_LIB=molecular-dynamics_amd/libmdx_ts.so timeout -k 10 300 python tools/step_stamps.py > $O/step_stamps.txt 2>&1 || exit 1
bash tools/nb_pmc.sh > $O/nb_pmc.log 2>&1 || { tail -5 $O/nb_pmc.log; exit 1; }
cp gpurun_out/nb_pmc_traffic.json $O/
bash tools/nb_sq3.sh > $O/nb_sq_counters.txt 2>&1 || exit 1
KERN="k_nonbonded_ev(" MDX_NB_ENERGY=1 bash tools/nb_sq3.sh > $O/nb_energy_sq_counters.txt 2>&1 || exit 1
timeout -k 10 300 python tools/plain_step_period.py > $O/plain_step_period.txt 2>/dev/null || exit 1
timeout -k 10 300 python tools/frame_cost.py 2>/dev/null | tail -2 > $O/frame_cost.txt || exit 1
bash tools/frame_trace.sh > /dev/null 2>&1 || exit 1
python - <<'PY'
import json
for f in ("bench_default", "bench_20_steps_driver_invocation"):
    d = json.loads(open("gpurun_out/r03/%s.json" % f).readline())
    print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("cli_ns_per_day"), d.get("cpu_baseline", {}).get("value"))
PY
