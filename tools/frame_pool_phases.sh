#!/bin/bash
# bash tools/frame_pool_phases.sh  -> gpurun_out/frame_pool_phases.txt: per-kernel averages of the pool's mix / tail phases, separate and fused
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/tools/frame_pool_phases.py 1 > /dev/null 2>&1
: > $R/gpurun_out/frame_pool_phases.txt
for mode in 2 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fpp_$mode -o p -- python $R/tools/frame_pool_phases.py $mode > /dev/null 2>&1
  f=$(ls $R/gpurun_out/fpp_$mode/*kernel_stats.csv | head -1)
  echo "mode $mode (2: mix and tail as two kernels; 1: fused)" >> $R/gpurun_out/frame_pool_phases.txt
  python - "$f" >> $R/gpurun_out/frame_pool_phases.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'pool' in r['Name']:
        print(f"  {r['Name'][:100]:100s} {r['Calls']:>5s} launches  avg {float(r['AverageNs']) / 1e3:7.2f} us  min {float(r['MinNs']) / 1e3:7.2f}  max {float(r['MaxNs']) / 1e3:7.2f}")
PY
  rm -rf $R/gpurun_out/fpp_$mode
done
cat $R/gpurun_out/frame_pool_phases.txt
