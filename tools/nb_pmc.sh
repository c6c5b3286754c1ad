#!/bin/bash
# HBM traffic counters of the force-only nonbonded kernel, one counter per pass (GPU box, repository root).
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o p -- python3 $R/tools/nb_pmc.py > $R/gpurun_out/pmc_$c.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/pmc_$c -name "*counter_collection.csv" | tail -1)
  python3 $R/tools/pmc_summary.py $f k_nonbonded_f3 $c
done
