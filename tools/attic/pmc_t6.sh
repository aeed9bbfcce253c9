#!/bin/bash
# PMC passes for the 16-bit 48-channel convolution launch (conv_t6.hip): tools/pmc_t6.sh <outdir> [LDS_TUNE codes]
mkdir -p $1; out=$(cd $1 && pwd); here=$(cd $(dirname $0)/.. && pwd)
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  LDS_TUNE=$2 timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o pmc -- python $here/tools/prof_kernel.py conv_bf16 6 > $out/p$i.log 2>&1
done
python $here/tools/pmc_summary.py $out
