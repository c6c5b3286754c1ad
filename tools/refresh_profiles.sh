set -e
R=$PWD
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_gpu_v3.log 2>&1
timeout -k 10 200 python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/smoke.log 2>&1
timeout -k 10 400 python bench.py > gpurun_out/bench_v7.json 2> gpurun_out/bench_v7.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v7 -o p -- python3 $R/bench.py --steps 2000 --warmup 300 --no-cpu-baseline > $R/gpurun_out/prof_v7.log 2>&1
cd $R
tail -2 gpurun_out/pytest_gpu_v3.log; cat gpurun_out/bench_v7.json
MDX_LIB=molecular-dynamics_amd/libmdx_ts.so timeout -k 10 300 python tools/step_stamps.py > gpurun_out/step_stamps_v7.txt 2>&1
bash tools/nb_pmc.sh > gpurun_out/nb_pmc.log 2>&1
bash tools/nb_sq.sh > gpurun_out/nb_sq_v7.txt 2>&1
