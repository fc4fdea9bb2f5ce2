#!/bin/bash
# Development aid (GPU box): PMC passes over one assembly of the bench cube with both assembly kernels -> gpurun_out/assembly_pmc.txt
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/assembly_pmc.txt
: > $OUT
for k in tets rows; do
  export FEMBRAIN_ASM_KERNEL=$k
  echo "== FEMBRAIN_ASM_KERNEL=$k" >> $OUT
  $R/tools/pmc_kernel.sh "FETCH_SIZE" "k_assemble" $R/tools/assemble_once.py >> $OUT
  $R/tools/pmc_kernel.sh "WRITE_SIZE" "k_assemble" $R/tools/assemble_once.py >> $OUT
  $R/tools/pmc_kernel.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" "k_assemble" $R/tools/assemble_once.py >> $OUT
  $R/tools/pmc_kernel.sh "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_INSTS_SALU" "k_assemble" $R/tools/assemble_once.py >> $OUT
done
unset FEMBRAIN_ASM_KERNEL
cat $OUT
