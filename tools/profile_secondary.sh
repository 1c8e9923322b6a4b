#!/bin/bash
# rocprofv3 kernel stats of the secondary workloads:  bash tools/profile_secondary.sh r03   -> gpurun_out/<tag>_cfg5_kernel_stats.csv, <tag>_train_kernel_stats.csv
set -x
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/tools/cfg5_profile_target.py > /dev/null 2>&1      # tile choices into the tuning cache first (no timing launches in the profile)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_cfg5_prof -o cfg5 -- python $R/tools/cfg5_profile_target.py > $R/gpurun_out/${TAG}_cfg5.log 2>&1
python $R/tools/train_step_time.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_train_prof -o train -- python $R/tools/train_step_time.py > $R/gpurun_out/${TAG}_train.log 2>&1
cd $R
for w in cfg5 train; do
  f=$(ls gpurun_out/${TAG}_${w}_prof/*kernel_stats.csv | head -1); cp $f gpurun_out/${TAG}_${w}_kernel_stats.csv; rm -rf gpurun_out/${TAG}_${w}_prof
  head -25 gpurun_out/${TAG}_${w}_kernel_stats.csv | cut -c1-200
done
grep "ms per" gpurun_out/${TAG}_cfg5.log gpurun_out/${TAG}_train.log
