"""A/B (round 6) of one d4_debug_switch of the bf16 engine on BASELINE config 5 at B = 128 and B = 1024, 6 frames, same process, alternating
(a fresh engine per run: a captured decode graph keeps the launch sequence it was recorded with).
    python tools/cfg5_switch_ab.py pool_wide_keys [B ...]       a hidden projected once for every later pool, against one key projection per pool"""
import sys
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
import bench
from dreamer4_amd import _lib

lib = _lib.load()
name = sys.argv[1].encode()
VALS = (1, 0)
if ':' in sys.argv[1]:                      # name:a,b  -> the two values to compare
    name = sys.argv[1].split(':')[0].encode(); VALS = tuple(int(v) for v in sys.argv[1].split(':')[1].split(','))
for B in [int(a) for a in sys.argv[2:]] or [128, 1024]:
    out = {}
    for rnd in range(2):
        for on in VALS:
            lib.d4_debug_switch(name, on)
            r = bench.cfg5_bf16('cuda', lib, B=B, frames=6, reps=2)
            rf = r['roofline']
            out.setdefault(on, []).append((r['ms_per_rollout'], rf['achieved'], rf['avg_launch_us'], rf['launches_timed']))
            torch.cuda.empty_cache()
    lib.d4_debug_switch(name, 1)
    for on in VALS:
        print(f"B={B:5d} on={on}: " + ' | '.join(f'{ms:8.2f} ms per 6-frame rollout, GEMMs {tf:6.1f} TF/s ({us:.1f} us x {n} timed)' for ms, tf, us, n in out[on]), flush=True)
