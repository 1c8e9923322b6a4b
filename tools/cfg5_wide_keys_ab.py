"""A/B (round 6): the bf16 engine's wide key projection (a hidden projected once for every later pool, d4_debug_switch 'pool_wide_keys') against one
key projection per pool, BASELINE config 5 at B = 128 and B = 1024, 6 frames, same process, alternating.    python tools/cfg5_wide_keys_ab.py [B ...]"""
import sys
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
import bench
from dreamer4_amd import _lib

lib = _lib.load()
for B in [int(a) for a in sys.argv[1:]] or [128, 1024]:
    out = {}
    for rnd in range(2):
        for wide in (1, 0):
            lib.d4_debug_switch(b'pool_wide_keys', wide)
            r = bench.cfg5_bf16('cuda', lib, B=B, frames=6, reps=2)
            rf = r['roofline']
            out.setdefault(wide, []).append((r['ms_per_rollout'], rf['achieved'], rf['avg_launch_us'], rf['launches_timed']))
            torch.cuda.empty_cache()
    lib.d4_debug_switch(b'pool_wide_keys', 1)
    for wide in (1, 0):
        print(f"B={B:5d} wide={wide}: " + ' | '.join(f'{ms:8.2f} ms per 6-frame rollout, GEMMs {tf:6.1f} TF/s ({us:.1f} us x {n} timed)' for ms, tf, us, n in out[wide]), flush=True)
