#!/bin/bash
# Is config 4's env step bound by the GPU or by the host?  Sum of kernel durations per env step (rocprofv3 --kernel-trace --stats) against the wall time
# the same script prints un-profiled.   bash tools/cfg4_gpu_busy.sh  -> gpurun_out/cfg4_gpu_busy.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/tools/cfg4_profile_target.py 1 > $R/gpurun_out/cfg4_wall.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cfg4p -o p -- python $R/tools/cfg4_profile_target.py 1 > $R/gpurun_out/cfg4_wall_profiled.txt 2>&1
cd $R
python - <<'PY' > gpurun_out/cfg4_gpu_busy.txt
import csv, glob
f = glob.glob('gpurun_out/cfg4p/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
steps = 60      # 2 passes x 30 env steps
print(open('gpurun_out/cfg4_wall.txt').read().strip().splitlines()[-1], '(un-profiled wall)')
print(open('gpurun_out/cfg4_wall_profiled.txt').read().strip().splitlines()[-1], '(wall under the profiler)')
print(f'kernel time summed: {tot / 1e6:.2f} ms over {calls} launches in the whole script (model set-up included) = at most {tot / 1e6 / steps:.3f} ms and {calls / steps:.0f} launches per env step')
for r in rows[:14]:
    print(f"  {r['Name'][:90]:90s} {int(r['Calls']) / steps:7.1f} / step  avg {float(r['AverageNs']) / 1e3:6.2f} us  {float(r['TotalDurationNs']) / 1e6 / steps * 1e3:7.1f} us / step")
PY
rm -rf gpurun_out/cfg4p
cat gpurun_out/cfg4_gpu_busy.txt
