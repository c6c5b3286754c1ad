#!/bin/bash
# Instruction and LDS accounting of a force-only nonbonded kernel (GPU box, repository root).
# usage: tools/nb_sq4.sh KERNEL_NAME [MDX_DEBUG switches]
R=$PWD
K=${1:-"k_nonbonded_ft("}
export MDX_DEBUG=$2
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
rm -rf $R/gpurun_out/pmc_sq4
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_sq4 -o p -- python3 $R/tools/nb_pmc.py > $R/gpurun_out/pmc_sq4.log 2>&1 || { tail -5 $R/gpurun_out/pmc_sq4.log; exit 1; }
f=$(find $R/gpurun_out/pmc_sq4 -name "*counter_collection.csv" | tail -1)
for c in $C; do python3 $R/tools/pmc_summary.py $f $K $c; done
