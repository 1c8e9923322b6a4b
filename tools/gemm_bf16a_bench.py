"""GPU microbenchmark: the bf16-activation LDS-DMA GEMM (gemm_bf16a.hip) against the register-staged bf16 GEMM that reads fp32 activations
(gemm_bf16.hip), on the config-5 shapes (dim 1024, B = 128 -> 1792 rows); correctness of every tile configuration against a float64
product of the bf16-rounded operands."""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ctypes as C
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
RMS, SWIGLU = 1, 4
D, hd, inner = 1024, 512, 2752
shapes = [(1792, 3 * hd + 16, D, RMS, 'proj'), (1792, D, hd, 0, 'out'), (1792, 2 * inner, D, RMS | SWIGLU, 'ff1'), (1792, D, inner, 0, 'ff2'),
          (1792, 256, D, RMS, 'poolq'), (1792 * 5, 256, D, RMS, 'poolk5'), (1792 * 13, 256, D, RMS, 'poolk13'), (1792 * 25, 256, D, RMS, 'poolk25'),
          (1792, D, 256, 0, 'poolout'), (4096, 4096, 4096, 0, 'cube4k')]
reps = 20
NC = 6
tot_old = tot_new = 0.
for M, N, K, flags, name in shapes:
    g = torch.Generator(device='cuda').manual_seed(1)
    A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    Ab, Wb = A.to(torch.bfloat16).contiguous(), W.to(torch.bfloat16).contiguous()
    b = torch.randn(N, device='cuda', generator=g)
    Nout = N // 2 if flags & SWIGLU else N
    out = torch.empty(M, Nout, device='cuda'); outb = torch.empty(M, Nout, device='cuda', dtype=torch.bfloat16)
    eps = 1e-6
    ref = None
    if M * N * K < 3e11:
        Ad = Ab.double()
        X = Ad * torch.rsqrt(Ad.pow(2).mean(-1, keepdim=True) + eps) if flags & RMS else Ad
        ref = X @ Wb.double().t() + b.double()
        if flags & SWIGLU:
            r = ref.reshape(M, N // 64, 2, 32)
            ref = (r[:, :, 0] * torch.nn.functional.silu(r[:, :, 1])).reshape(M, N // 2)

    def timeit(run):
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    lib.d4_gemm_force_config(-1)
    t_old = timeit(lambda: lib.d4_gemm_bf16(_lib.ptr(A), K, _lib.ptr(Wb), K, _lib.ptr(out), Nout, _lib.ptr(b), None, 0, M, N, K, flags, eps, s))
    ts, errs = [], []
    for c in range(NC):
        out.fill_(float('nan'))
        rc = lib.d4_gemm_bf16a(_lib.ptr(Ab), K, _lib.ptr(Wb), K, _lib.ptr(out), Nout, _lib.ptr(outb), _lib.ptr(b), None, 0, M, N, K, flags, eps, c, s)
        if rc != 0:
            ts.append(float('nan')); continue
        t = timeit(lambda: lib.d4_gemm_bf16a(_lib.ptr(Ab), K, _lib.ptr(Wb), K, _lib.ptr(out), Nout, _lib.ptr(outb), _lib.ptr(b), None, 0, M, N, K, flags, eps, c, s))
        ts.append(t)
        if ref is not None:
            errs.append(((out.double() - ref).abs().max() / ref.abs().max()).item())
            assert torch.equal(outb, out.to(torch.bfloat16)), 'bf16 copy differs from the rounded fp32 output'
    t_rule = timeit(lambda: lib.d4_gemm_bf16a(_lib.ptr(Ab), K, _lib.ptr(Wb), K, _lib.ptr(out), Nout, _lib.ptr(outb), _lib.ptr(b), None, 0, M, N, K, flags, eps, -1, s))
    fl = 2.0 * M * N * K
    best = min(t for t in ts if t == t)
    tot_old += t_old; tot_new += t_rule
    print(f'{name:8s} M{M:6d} N{N:5d} K{K:5d} f{flags}: fp32-A kernel {t_old:7.1f} us {fl / t_old / 1e6:6.0f} TF/s | bf16-A per cfg ' + ' '.join(f'{t:7.1f}' for t in ts) +
          f' | rule {t_rule:7.1f} us {fl / t_rule / 1e6:6.0f} TF/s (best {fl / best / 1e6:6.0f}) | max rel err {max(errs) if errs else float("nan"):.1e}')
print('configs: 128x128 128x64 64x64 64x64/s 32x64 256x128 ; sum fp32-A', round(tot_old), 'us, bf16-A by rule', round(tot_new), 'us')
