"""In-situ refinement of the shipped tile table: for the rollout's heaviest fp32 GEMM shapes, try the other configurations of the LDS-DMA family with the
WHOLE cfg-2 rollout as the clock (a shape's best tile in isolation is not always its best between its neighbours: FF2 32x32 vs 32x64: -1.1 ms per rollout).
Every configuration of a family gives the same bits, so this changes time only.   python tools/insitu_tune.py  -> gpurun_out/insitu_tune.txt"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = open(os.path.join(ROOT, 'dreamer4_amd', 'gemm_tune_default.txt')).read().splitlines()
keys = ['3584 512 1376 0 1', '3584 1552 512 1 1', '10752 256 512 1 1', '17920 256 512 1 1', '25088 256 512 1 1', '32256 256 512 1 1', '39424 256 512 1 1',
        '13312 256 512 1 1', '3840 512 1376 0 1', '3840 1552 512 1 1']
cands = [100, 102, 104, 106, 108]

def run(lines):
    path = os.path.join(ROOT, 'gpurun_out', 'insitu_table.txt')
    open(path, 'w').write('\n'.join(lines) + '\n')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rollout_time.py')], capture_output=True, text=True, env=dict(os.environ, D4_GEMM_TUNE_DEFAULT=path))
    return float(out.stdout.strip().split('min ')[1].rstrip(')'))

cur = list(base)
best = run(cur)
log = [f'baseline {best:.2f} ms']
print(log[-1], flush=True)
for k in keys:
    idx = [i for i, l in enumerate(cur) if l.startswith(k + ' ')]
    if not idx:
        log.append(f'{k}: not in the table'); print(log[-1], flush=True); continue
    i = idx[0]
    now = int(cur[i].split()[-1])
    for c in cands:
        if c == now:
            continue
        trial = list(cur); trial[i] = f'{k} {c}'
        t = run(trial)
        log.append(f'{k}: {now} -> {c}: {t:.2f} ms (best {best:.2f})'); print(log[-1], flush=True)
        if t < best - 0.25:
            best, cur, now = t, trial, c
open(os.path.join(ROOT, 'gpurun_out', 'insitu_tune.txt'), 'w').write('\n'.join(log) + '\n')
open(os.path.join(ROOT, 'gpurun_out', 'insitu_table_best.txt'), 'w').write('\n'.join(cur) + '\n')
print('final', best)
