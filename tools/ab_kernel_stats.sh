#!/bin/bash
# Per-kernel averages of the bench under rocprofv3 for two builds of the library (GPU box, repository root):
#   bash tools/ab_kernel_stats.sh LIB_A.so LIB_B.so
R=$PWD
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  tag=$(basename $L .so)
  rm -rf $R/gpurun_out/st_$tag
  MDX_LIB=$R/molecular-dynamics_amd/$L timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_$tag -o s -- python3 $R/bench.py --no-cpu-baseline --no-cli --steps 2000 --warmup 100 > $R/gpurun_out/st_$tag.log 2>&1 || { tail -5 $R/gpurun_out/st_$tag.log; exit 1; }
  f=$(find $R/gpurun_out/st_$tag -name "*kernel_stats.csv" | tail -1)
  echo "== $tag"
  python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s %6s calls  avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
