"""BASELINE config 5's rollout on the bf16 path at the per-GPU batch the config asks for (128 = 1024 / 8) and at larger batches on ONE GPU:
how much of the 0.087-of-peak figure is the 1792-row problem and how much the kernels.    python tools/cfg5_batch_sweep.py [B ...]"""
import sys
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
import bench
from dreamer4_amd import _lib

lib = _lib.load()
for B in [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024]:
    r = bench.cfg5_bf16('cuda', lib, B=B, frames=6, reps=1)
    rf = r['roofline']
    print(f"B={B:5d} ({B * 14} token rows per evaluation): {r['value']:9.1f} imagined steps/s, {r['ms_per_rollout']:8.1f} ms per 6-frame rollout, bf16 GEMMs "
          f"{rf['achieved']:7.1f} TF/s = {rf['frac']:.3f} of the bf16 peak, {rf['avg_launch_us']:.1f} us per launch, algorithmic {r['rollout_algorithmic_tflops']:.0f} TF/s", flush=True)
    torch.cuda.empty_cache()
