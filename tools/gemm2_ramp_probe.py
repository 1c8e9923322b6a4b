"""Round 6, VERDICT r5 #4: how much of a launch of the dominant fp32 class (gemm2_kernel<2,2,1,2,3,32>: 32 x 64 tiles, 16 waves per CU) sits OUTSIDE its k-loop —
the only part a strip-persistent form (A panel resident / the DMA ring running on across the column tiles of a strip) could remove.  For the class's shapes at
cfg 2 the same kernel is timed with the contraction K in {256, 512, 1024, 2048} (same M, N, tile, epilogue, grid): t(K) = a + b K; `a` is the launch's fixed part
(grid ramp, first-tile latency, epilogue, tail round), b * 512 the k-loop of the real call.    python tools/gemm2_ramp_probe.py"""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ctypes as C
import statistics
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
RMS = 1
CFG_32x64 = 100 + 6
shapes = [(3584, 1552, RMS, False, 'fused q|k|v projection'), (3584, 512, 0, True, 'output projection (+ residual)'), (25088, 256, RMS, False, 'pool keys, L = 7'),
          (39424, 256, RMS, False, 'pool keys, L = 11')]
Ks = (256, 512, 1024, 2048)


def timed(run, inner=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner * 1e3


lib.d4_gemm_force_config(CFG_32x64)
try:
    for M, N, flags, resid, name in shapes:
        g = torch.Generator(device='cuda').manual_seed(1)
        runs = {}
        keep = []
        for K in Ks:
            A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
            out = torch.empty(M, N, device='cuda'); R = torch.randn(M, N, device='cuda', generator=g) if resid else None
            keep.append((A, W, out, R))
            runs[K] = (lambda A=A, W=W, out=out, R=R, K=K: lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(out), N, None, _lib.ptr(R), N, M, N, K, flags, 1e-6, s))
            assert runs[K]() == 0
        ts = {K: [] for K in Ks}
        for _ in range(5):
            for K in Ks:
                ts[K].append(timed(runs[K]))
        t = {K: statistics.median(v) for K, v in ts.items()}
        # least squares t = a + b K over the four contractions
        n = len(Ks); sx = sum(Ks); sy = sum(t.values()); sxx = sum(k * k for k in Ks); sxy = sum(k * t[k] for k in Ks)
        b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
        fl = 2.0 * M * N * 512
        print(f'{name:32s} M{M:6d} N{N:5d}: ' + '  '.join(f'K={K}: {t[K]:6.1f} us' for K in Ks) +
              f' | fixed part a = {a:5.1f} us ({100 * a / t[512]:4.1f} % of the K = 512 call), k-loop {b * 512:5.1f} us = {fl / (b * 512) / 1e6:5.1f} TF/s, whole call {fl / t[512] / 1e6:5.1f} TF/s', flush=True)
finally:
    lib.d4_gemm_force_config(-1)
