#!/bin/bash
# Kernel + memory-copy trace of a short production bench with trajectory output (GPU box, repository root).
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/trace_out -o t -- python3 $R/bench.py --steps 400 --warmup 100 --no-cpu-baseline > $R/gpurun_out/trace_out.log 2>&1 || exit 1
cd $R
ls gpurun_out/trace_out
