#!/bin/bash
# L2 (TCC) / L1 (TCP) counter passes over the cfg-2 rollout:  bash tools/l2_pmc.sh r03a  -> gpurun_out/<tag>_l2_pmc.txt
# Counter passes are separate rocprofv3 runs with --kernel-trace only (TCC has 4 slots per pass).
set -x
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
export D4_GEMM_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/tune_cache_$TAG.txt
[ -f $D4_GEMM_TUNE_CACHE ] || python tools/rollout_profile_target.py      # fills the tuning cache so that no timing launches are counted
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum GRBM_GUI_ACTIVE" \
            "TCC_TAG_STALL_sum TCC_READ_sum TCC_WRITE_sum TCC_CYCLE_sum GRBM_GUI_ACTIVE" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/l2pmc_$i -o p -- python $GRAFT_REPO_ROOT/tools/rollout_profile_target.py > $GRAFT_REPO_ROOT/gpurun_out/l2pmc_$i.log 2>&1
  tail -3 $GRAFT_REPO_ROOT/gpurun_out/l2pmc_$i.log
done
cd $GRAFT_REPO_ROOT
dirs=""; for d in gpurun_out/l2pmc_1 gpurun_out/l2pmc_2 gpurun_out/l2pmc_3; do ls $d/*counter_collection.csv > /dev/null 2>&1 && dirs="$dirs $d"; done
python tools/pmc_summary.py $dirs > gpurun_out/${TAG}_l2_pmc.txt
rm -rf gpurun_out/l2pmc_1 gpurun_out/l2pmc_2 gpurun_out/l2pmc_3
head -12 gpurun_out/${TAG}_l2_pmc.txt | cut -c1-400
