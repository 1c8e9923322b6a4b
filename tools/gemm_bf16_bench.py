"""GPU microbenchmark: the bf16 MFMA GEMM on the config-5 shapes (dim 1024, B = 128 -> 1792 / 1920 rows), per tile configuration."""
import sys; sys.path.insert(0, '/root/repo')
import ctypes as C
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
RMS, SWIGLU = 1, 4
D, hd, inner = 1024, 512, 2752
shapes = [(1792, 3 * hd + 16, D, RMS, 'proj'), (1792, D, hd, 0, 'out'), (1792, 2 * inner, D, RMS | SWIGLU, 'ff1'), (1792, D, inner, 0, 'ff2'),
          (1792, 256, D, RMS, 'poolq'), (1792 * 5, 256, D, RMS, 'poolk5'), (1792 * 13, 256, D, RMS, 'poolk13'), (1792 * 25, 256, D, RMS, 'poolk25'),
          (1792, D, 256, 0, 'poolout'), (128 * 64, 2 * hd, 32, RMS, 'lkv'), (128 * 64, 32, hd, 0, 'lout')]
reps = 20
cfgs = [0, 1, 2, 3, 4, 5, 100, 101, 102, 103, 104]          # register-staged forms, then the LDS-DMA forms (gemm_bf16_dma.hip)
ncfg = len(cfgs)
tot = [0.] * ncfg; best_tot = 0.
for M, N, K, flags, name in shapes:
    g = torch.Generator(device='cuda').manual_seed(1)
    A = torch.randn(M, K, device='cuda', generator=g); Wb = torch.randn(N, K, device='cuda', generator=g).to(torch.bfloat16).contiguous()
    b = torch.randn(N, device='cuda', generator=g)
    Nout = N // 2 if flags & SWIGLU else N
    out = torch.empty(M, Nout, device='cuda')
    ts = []; err = 0.
    ref = None
    for c in cfgs:
        lib.d4_gemm_force_config(200 + c)
        run = lambda: lib.d4_gemm_bf16(_lib.ptr(A), K, _lib.ptr(Wb), K, _lib.ptr(out), Nout, _lib.ptr(b), None, 0, M, N, K, flags, 1e-6, s)
        for _ in range(3): run()
        torch.cuda.synchronize()
        if c == 0: ref = out.clone()
        elif c >= 100: err = max(err if c > 100 else 0., (out - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    lib.d4_gemm_force_config(-1)
    fl = 2.0 * M * N * K
    best_tot += min(ts)
    print(f'{name:8s} M{M:6d} N{N:5d} K{K:5d} f{flags}: ' + ' '.join(f'{t:7.1f}' for t in ts) + f' us | best {fl / min(ts) / 1e6:7.1f} TF/s | dma vs staged max rel diff {err:.1e}')
print('configs: staged 128x128/4w 128x128/8w 64x128 64x64 256x128/8w 128x64 | dma 128x128 128x256 64x128 256x128/8w 128x128/8w ; sum of best', round(best_tot), 'us')
