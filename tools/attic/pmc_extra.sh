#!/bin/bash
# extra memory-path counters for one kernel (see tools/pmc_run.sh)
what=$1; out=$2; mkdir -p $out; here=$(cd $(dirname $0)/.. && pwd)
cd /tmp; export TMPDIR=/tmp
i=10
for set in "GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY" "TCP_PENDING_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCC_BUSY_avr TCC_REQ_sum TCC_TAG_STALL_sum TCC_READ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o pmc -- python $here/tools/prof_kernel.py $what 6 > $out/p$i.log 2>&1
done
python $here/tools/pmc_summary.py $out
