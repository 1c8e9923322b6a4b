#!/bin/bash
# Round measurements on the GPU box:  bash tools/measure_round.sh r02a   -> gpurun_out/<tag>_*  (copy the summaries into profiles/)
set -x
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
export D4_GEMM_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/tune_cache_$TAG.txt
rm -f $D4_GEMM_TUNE_CACHE
# 1. plain bench (the driver's command), then the driver-style torchrun launch with the 1-rank RCCL group forced
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
D4_FORCE_PG=1 NCCL_DEBUG=INFO python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_rccl1.json 2> gpurun_out/${TAG}_bench_rccl1.err
cat gpurun_out/${TAG}_bench_rccl1.err gpurun_out/${TAG}_bench_rccl1.json | grep -i "NCCL INFO" | grep -i "init\|comm\|version\|Using\|Channel" | head -30 > gpurun_out/${TAG}_rccl_init.log
# 2. kernel trace + stats of the same command
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_profiled.json 2> /dev/null
rm -f $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof/bench_kernel_trace.csv
# 3. PMC passes over a short rollout (no tuning launches: cache present)
for pass in "FETCH_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "WRITE_SIZE SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o p -- python $GRAFT_REPO_ROOT/tools/rollout_profile_target.py > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE --json gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_rollout_pmc.txt
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
cp $D4_GEMM_TUNE_CACHE gpurun_out/${TAG}_gemm_tile_choices.txt
head -8 gpurun_out/${TAG}_rollout_pmc.txt | cut -c1-220
cut -c1-700 gpurun_out/${TAG}_bench.json
tail -3 gpurun_out/${TAG}_bench_rccl1.err; cut -c1-300 gpurun_out/${TAG}_bench_rccl1.json; head -5 gpurun_out/${TAG}_rccl_init.log
