#!/bin/bash
# L2 / fabric counters of one gemm_bf16a launch shape/config: bash tools/bf16a_pmc2.sh ff1 0
SH=${1:-ff1}; CFG=${2:-0}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmcc_$i -o p -- python $R/tools/bf16a_pmc_target.py $SH $CFG > /dev/null 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/pmcc_1 gpurun_out/pmcc_2 gpurun_out/pmcc_3 2>&1 | grep -i "kernel\|bf16a" > gpurun_out/pmc2_bf16a_${SH}_${CFG}.txt
rm -rf gpurun_out/pmcc_*
cat gpurun_out/pmc2_bf16a_${SH}_${CFG}.txt
