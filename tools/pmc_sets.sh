#!/bin/bash
# PMC counters of ONE kernel in separate rocprofv3 passes (MI355X_MICROARCH.md: 8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE
# in passes of their own).  usage: tools/pmc_sets.sh <prof_kernel.py workload> <outdir> [full]   (env LDS_TUNE etc. pass through)
what=$1; mkdir -p $2; out=$(cd $2 && pwd); here=$(cd $(dirname $0)/.. && pwd)
cd /tmp; export TMPDIR=/tmp
i=0
sets=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
      "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC TCC_HIT_sum TCC_MISS_sum")
if [ "$3" = "full" ]; then sets+=("FETCH_SIZE" "WRITE_SIZE"); fi
for set in "${sets[@]}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o pmc -- python $here/tools/prof_kernel.py $what 6 > $out/p$i.log 2>&1
done
python $here/tools/pmc_summary.py $out
