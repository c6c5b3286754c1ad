#!/bin/bash
# Kernel trace of a short production bench (run on the GPU box from the repository root); prints the
# timeline of a plain step, of an energy/virial step and of a search step.
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -o t -- python3 $R/bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-output "$@" > $R/gpurun_out/trace.log 2>&1 || exit 1
cd $R
f=$(find gpurun_out/trace -name "*kernel_trace.csv" | tail -1)
python3 tools/trace_step.py $f 203 > gpurun_out/step_plain.txt
python3 tools/trace_step.py $f 210 > gpurun_out/step_energy.txt
python3 tools/trace_step.py $f 220 > gpurun_out/step_search.txt
head -1 gpurun_out/step_plain.txt gpurun_out/step_energy.txt gpurun_out/step_search.txt
