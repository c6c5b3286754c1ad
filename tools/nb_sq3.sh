#!/bin/bash
# Wave-cycle accounting of the force-only nonbonded kernel in one counter pass (GPU box, repository root).
R=$PWD
cd /tmp && export TMPDIR=/tmp
KERN=${KERN:-"k_nonbonded_ft("}
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_sq3 -o p -- python3 $R/tools/nb_pmc.py > $R/gpurun_out/pmc_sq3.log 2>&1 || { tail -5 $R/gpurun_out/pmc_sq3.log; exit 1; }
f=$(find $R/gpurun_out/pmc_sq3 -name "*counter_collection.csv" | tail -1)
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU; do python3 $R/tools/pmc_summary.py $f "$KERN" $c; done
