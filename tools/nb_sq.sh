#!/bin/bash
# Wave-cycle accounting of the force-only nonbonded kernel (GPU box, repository root).
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_sq -o p -- python3 $R/tools/nb_pmc.py > $R/gpurun_out/pmc_sq.log 2>&1 || { tail -5 $R/gpurun_out/pmc_sq.log; exit 1; }
f=$(find $R/gpurun_out/pmc_sq -name "*counter_collection.csv" | tail -1)
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU; do python3 $R/tools/pmc_summary.py $f "k_nonbonded_ft(" $c; done
