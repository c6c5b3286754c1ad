#!/bin/bash
# A/B of engine switches on one box: bash tools/ab_bench.sh "label=KEYS" ... (KEYS as for MDX_DEBUG, "-" for none)
for spec in "$@"; do
  label=${spec%%=*}; keys=${spec#*=}; [ "$keys" = "-" ] && keys=""
  MDX_DEBUG="$keys" timeout -k 10 200 python bench.py --no-cpu-baseline --no-cli --steps 3000 --warmup 300 > gpurun_out/ab_$label.json 2> gpurun_out/ab_$label.err || { tail -3 gpurun_out/ab_$label.err; exit 1; }
  python - "$label" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_%s.json" % sys.argv[1]))
print("%-28s %8.1f ns/day  %.4f ms/step  nb %.1f us" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"] * 1e3))
PY
done
