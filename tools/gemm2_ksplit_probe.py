"""Round 6: the few-row / long-K form (gemm2_ksplit_kernel: K cut four ways inside a 16-wave workgroup) against the family's tiled configurations on the heads' shapes.
    python tools/gemm2_ksplit_probe.py"""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ctypes as C
import statistics
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
SILU = 2
shapes = [(256, 2048, 2048, SILU, 'cfg2 head hidden layer'), (256, 2048, 2048, 0, 'cfg2 head hidden layer, plain'), (128, 4096, 4096, SILU, 'cfg5 head hidden layer'),
          (256, 2048, 1024, SILU, 'K = 1024'), (512, 2048, 2048, SILU, '512 rows'), (1024, 512, 1376, 0, 'final-stage FF out (compact rows)'), (1024, 512, 512, 0, 'compact-row projection'),
          (1024, 1024, 512, 0, 'compact 1024'), (3584, 512, 512, 0, 'time-layer output projection'), (256, 2048, 512, SILU, 'head first layer'), (8192, 32, 512, 0, 'latent head')]
n2 = sum(lib.d4_profile_class_name(c).decode().startswith('gemm2_kernel') for c in range(lib.d4_profile_classes()))


def timed(run, inner=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner * 1e3


for M, N, K, flags, name in shapes:
    g = torch.Generator(device='cuda').manual_seed(1)
    A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5; b = torch.randn(N, device='cuda', generator=g)
    out = torch.empty(M, N, device='cuda')
    run = lambda: lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(out), N, _lib.ptr(b), None, 0, M, N, K, flags, 1e-6, s)
    res = {}
    for cfg in [199] + [100 + c for c in range(n2)]:
        lib.d4_gemm_force_config(cfg)
        if run() != 0:
            continue
        run(); torch.cuda.synchronize()
        res[cfg] = statistics.median(timed(run) for _ in range(5))
    lib.d4_gemm_force_config(-1)
    best = min((t, c) for c, t in res.items() if c != 199)
    if 199 not in res:
        print(name, 'k-split form not applicable', res); continue
    print(f'{name:32s} M{M:4d} N{N:5d} K{K:5d}: k-split form {res[199]:6.1f} us = {2.0 * M * N * K / res[199] / 1e6:5.1f} TF/s | best tiled configuration {best[1] - 100} {best[0]:6.1f} us | all: ' +
          ' '.join(f'{res[c]:.1f}' for c in sorted(res) if c != 199), flush=True)
