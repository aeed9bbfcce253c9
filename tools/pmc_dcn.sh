#!/bin/bash
# usage: pmc_dcn.sh <tag> <DCN_TUNE>
here=/root/repo; out=$here/gpurun_out/pmc_$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; export DCN_TUNE=$2
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o pmc -- python $here/tools/prof_kernel.py dcn_f32 6 > $out/p$i.log 2>&1
done
python $here/tools/pmc_summary.py $out > $here/gpurun_out/pmc_$1.txt 2>&1
rm -rf $out
