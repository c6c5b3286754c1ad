set -e
R=$PWD
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_gpu_v3.log 2>&1
timeout -k 10 400 python bench.py > gpurun_out/bench_v3.json 2> gpurun_out/bench_v3.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v3 -o p -- python3 $R/bench.py --steps 2000 --warmup 300 --no-cpu-baseline > $R/gpurun_out/prof_v3.log 2>&1
cd $R
tail -2 gpurun_out/pytest_gpu_v3.log; cat gpurun_out/bench_v3.json
