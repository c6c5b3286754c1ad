#!/bin/bash
# Kernel and copy trace of a short run with a frame captured every 10 steps (tools/frame_cost.py, variant b).
# Usage (GPU box): bash tools/frame_trace.sh ; the timeline around one capture is printed by tools/frame_trace_show.py
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp FRAME_COST_TRACE=1
mkdir -p gpurun_out/r03/frame_trace
timeout -s KILL 280 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/frame_trace -o t -- python3 tools/frame_cost.py > gpurun_out/r03/frame_trace/run.log 2>&1
python3 tools/frame_trace_show.py gpurun_out/r03/frame_trace > gpurun_out/r03/frame_trace_timeline.txt
python3 tools/frame_trace_steps.py gpurun_out/r03/frame_trace > gpurun_out/r03/frame_trace_steps.txt
rm -f gpurun_out/r03/frame_trace/*/*_agent_info.csv
tail -80 gpurun_out/r03/frame_trace_timeline.txt
