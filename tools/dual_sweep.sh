#!/bin/bash
# Bench sweep over (outer lifetime, prune interval) of the dual pair list; arguments: pairs "nstlist:nstprune".
for cfg in "$@"; do
  IFS=: read a b c <<< "$cfg"; c=${c:-0}
  timeout -k 10 200 python bench.py --no-cpu-baseline --nstlist $a --nstprune $b --nb-chunk $c > gpurun_out/bench_dual_${a}_${b}.json 2> gpurun_out/bench_dual_${a}_${b}.err || exit 1
  python - gpurun_out/bench_dual_${a}_${b}.json $c <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], "chunk", sys.argv[2], "ns/day", d["value"], "ms/step", d["ms_per_step"], "nb ms", d["roofline"]["avg_launch_ms"], "rlist", d["config"]["rlist_nm"], d["config"].get("rlist_inner_nm"))
PY
done
