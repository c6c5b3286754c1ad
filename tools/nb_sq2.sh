#!/bin/bash
# Instruction mix and stall accounting of the force-only nonbonded kernel: three counter passes (GPU box, repository root).
R=$PWD
cd /tmp && export TMPDIR=/tmp
KERN=${KERN:-"k_nonbonded_ft("}
pass() {
  tag=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- python3 $R/tools/nb_pmc.py > $R/gpurun_out/pmc_$tag.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$tag.log; exit 1; }
  f=$(find $R/gpurun_out/pmc_$tag -name "*counter_collection.csv" | tail -1)
  for c in "$@"; do python3 $R/tools/pmc_summary.py $f "$KERN" $c; done
}
pass sqa SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 || exit 1
pass sqb SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR || exit 1
pass sqc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_LDS_ADDR_CONFLICT SQ_INSTS SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES || exit 1
