#!/bin/bash
# Kernel-level profile of the pair search (run on the GPU box from the repository root).
export PYTHONPATH=$PWD
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_search -o s -- python3 $R/tools/stage_bench.py > $R/gpurun_out/prof_search.log 2>&1 || exit 1
grep -a "^search" $R/gpurun_out/prof_search.log
