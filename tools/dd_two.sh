# usage: dd_two.sh LABEL SCRIPT  (env MDX_DEBUG passed through): two processes of SCRIPT at once on the one GPU
(WITH_GLOO=1 FAKE_RANK=0 timeout -k 5 ${TLIMIT:-60} python -u $2 9.0 > gpurun_out/two0_$1.log 2>&1 &)
WITH_GLOO=1 FAKE_RANK=1 timeout -k 5 ${TLIMIT:-60} python -u $2 9.0 > gpurun_out/two1_$1.log 2>&1
sleep 2
echo "== $1: $(tail -n 1 gpurun_out/two0_$1.log | cut -c1-60) | $(tail -n 1 gpurun_out/two1_$1.log | cut -c1-60)"
