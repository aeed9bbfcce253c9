#!/bin/bash
# PMC counters for ONE kernel, collected as MI355X_MICROARCH.md prescribes: separate rocprofv3 --pmc passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only.   usage: tools/pmc_run.sh <what> <outdir>
what=$1; mkdir -p $2; out=$(cd $2 && pwd); here=$(cd $(dirname $0)/.. && pwd)
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o pmc -- python $here/tools/prof_kernel.py $what 6 > $out/p$i.log 2>&1
done
python $here/tools/pmc_summary.py $out
