#!/bin/bash
# L2 hit rate per kernel of the config-5 rollout (B = 128 or $2):  bash tools/cfg5_l2_pmc.sh r06 128  -> gpurun_out/<tag>_cfg5_l2_b<B>.txt
TAG=${1:-r06}; B=${2:-128}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/tools/cfg5_profile_target.py bf16 $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/l2pmc -o p -- python $R/tools/cfg5_profile_target.py bf16 $B > /dev/null 2>&1
cd $R
python - <<PY > gpurun_out/${TAG}_cfg5_l2_b${B}.txt
import csv, glob, collections
f = glob.glob('gpurun_out/l2pmc/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:90]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'], k)
    if key not in seen:
        seen.add(key); n[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get('TCC_REQ_sum', 0))
print(f"{'kernel':92s} {'launches':>8s} {'L2 req / launch':>16s} {'hit rate':>9s}")
for k, c in rows[:24]:
    h, m = c.get('TCC_HIT_sum', 0), c.get('TCC_MISS_sum', 0)
    print(f"{k:92s} {n[k]:8d} {c.get('TCC_REQ_sum', 0) / max(n[k], 1):16.0f} {h / max(h + m, 1):9.3f}")
PY
rm -rf gpurun_out/l2pmc
cat gpurun_out/${TAG}_cfg5_l2_b${B}.txt | cut -c1-140
