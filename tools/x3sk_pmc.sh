#!/bin/bash
# SQ counters of the persistent split-operand GEMM on one shape: bash tools/x3sk_pmc.sh M N K flags   -> gpurun_out/pmc_x3sk_<M>x<N>x<K>.txt
M=${1:-3584}; N=${2:-2752}; K=${3:-512}; F=${4:-5}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmcx_$i -o p -- python $R/tools/x3_profile_target.py $M $N $K $F 7 > /dev/null 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/pmcx_1 gpurun_out/pmcx_2 gpurun_out/pmcx_3 gpurun_out/pmcx_4 gpurun_out/pmcx_5 2>&1 | grep -i "kernel\|x3sk" > gpurun_out/pmc_x3sk_${M}x${N}x${K}.txt
rm -rf gpurun_out/pmcx_*
cat gpurun_out/pmc_x3sk_${M}x${N}x${K}.txt
