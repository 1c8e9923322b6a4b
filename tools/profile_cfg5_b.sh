#!/bin/bash
# rocprofv3 kernel stats of the config-5 rollout at a given batch:  bash tools/profile_cfg5_b.sh r06 1024   -> gpurun_out/<tag>_cfg5_b<B>_kernel_stats.csv
TAG=${1:-r06}; B=${2:-1024}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/tools/cfg5_profile_target.py bf16 $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_cfg5b_prof -o cfg5 -- python $R/tools/cfg5_profile_target.py bf16 $B > $R/gpurun_out/${TAG}_cfg5_b${B}.log 2>&1
cd $R
f=$(ls gpurun_out/${TAG}_cfg5b_prof/*kernel_stats.csv | head -1); cp $f gpurun_out/${TAG}_cfg5_b${B}_kernel_stats.csv; rm -rf gpurun_out/${TAG}_cfg5b_prof
head -22 gpurun_out/${TAG}_cfg5_b${B}_kernel_stats.csv | cut -c1-180
grep "ms per" gpurun_out/${TAG}_cfg5_b${B}.log
