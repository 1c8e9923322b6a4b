#!/bin/bash
# PMC passes over one gemm_bf16a launch shape/config: bash tools/bf16a_pmc.sh ff1 0   -> gpurun_out/pmc_bf16a_<shape>_<cfg>.txt
SH=${1:-ff1}; CFG=${2:-0}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmcb_$i -o p -- python $R/tools/bf16a_pmc_target.py $SH $CFG > /dev/null 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/pmcb_1 gpurun_out/pmcb_2 gpurun_out/pmcb_3 gpurun_out/pmcb_4 gpurun_out/pmcb_5 2>&1 | grep -i "kernel\|bf16" > gpurun_out/pmc_bf16a_${SH}_${CFG}.txt
rm -rf gpurun_out/pmcb_*
cat gpurun_out/pmc_bf16a_${SH}_${CFG}.txt
