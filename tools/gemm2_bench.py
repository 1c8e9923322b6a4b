"""GPU microbenchmark: the two fp32 GEMM families on the engine's cfg-2 shapes (per configuration), plus an fp64 check.
    python tools/gemm2_bench.py [reps]"""
import sys; sys.path.insert(0, '/root/repo')
import ctypes as C
import torch
from dreamer4_amd import _lib

lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
RMS, SILU, SWIGLU = 1, 2, 4
shapes = [(3584, 1552, 512, RMS, 1, 'proj'), (3584, 2064, 512, RMS, 1, 'proj0'), (3584, 512, 512, 0, 1, 'out'), (3584, 2752, 512, RMS | SWIGLU, 1, 'ff1'),
          (3584, 512, 1376, 0, 1, 'ff2'), (3584, 256, 512, RMS, 1, 'poolq'), (10752, 256, 512, RMS, 1, 'poolk3'), (25088, 256, 512, RMS, 1, 'poolk7'),
          (39424, 256, 512, RMS, 1, 'poolk11'), (3584, 64, 512, 0, 4, 'poolv'), (3584, 512, 256, 0, 1, 'poolout'), (3840, 2752, 512, RMS | SWIGLU, 1, 'ff1c'),
          (3840, 512, 512, 0, 1, 'outc'), (1024, 512, 512, 0, 1, 'c_out'), (1024, 2752, 512, RMS | SWIGLU, 1, 'c_ff1'), (13312, 256, 512, RMS, 1, 'c_poolk'),
          (8192, 1024, 32, RMS, 1, 'lkv'), (8192, 32, 512, 0, 1, 'lout'), (4096, 2048, 2048, 0, 1, 'headL')]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n1 = lib.d4_gemm_force_config(-1)
n2 = sum(lib.d4_profile_class_name(c).decode().startswith('gemm2_kernel') for c in range(lib.d4_profile_classes()))


def bench(M, N, K, flags, batch, cfg, check):
    g = torch.Generator(device='cuda').manual_seed(1)
    if batch > 1:      # the pool's per-head value projection: A [M][batch*K] (head slices), W [batch*N][K], C [M][batch*N]
        A = torch.randn(M, batch * K, device='cuda', generator=g); W = torch.randn(batch * N, K, device='cuda', generator=g)
        Cc = torch.full((M, batch * N), float('nan'), device='cuda')
        args = (_lib.ptr(A), batch * K, _lib.ptr(W), K, _lib.ptr(Cc), batch * N)
    else:
        A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g)
        Nout = N // 2 if flags & SWIGLU else N
        Cc = torch.full((M, Nout), float('nan'), device='cuda')
        args = (_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(Cc), Nout)
    b = torch.randn(N, device='cuda', generator=g)
    R = torch.randn(M, N, device='cuda', generator=g) if not (flags & SWIGLU) and batch == 1 else None
    lib.d4_gemm_force_config(cfg)

    def run():
        if batch > 1:
            return lib.d4_gemm_batched(*args, None, None, 0, M, N, K, flags, 1.1920929e-07, batch, K, N * K, N, s)
        return lib.d4_gemm(*args, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, 1.1920929e-07, s)
    try:
        rc = run()
        if rc != 0:
            return None, None
        torch.cuda.synchronize()
        err = None
        if check and batch == 1:
            Ad, Wd = A.double(), W.double()
            X = Ad * torch.rsqrt(Ad.pow(2).mean(-1, keepdim=True) + 1.1920929e-07) if flags & RMS else Ad
            ref = X @ Wd.t() + b.double()
            if flags & SWIGLU:
                r = ref.reshape(M, N // 64, 2, 32)
                ref = (r[:, :, 0] * torch.nn.functional.silu(r[:, :, 1])).reshape(M, N // 2)
            if R is not None:
                ref = ref + R.double()
            err = (Cc.double() - ref).abs().max().item() / max(1., ref.abs().max().item())
        for _ in range(3):
            run()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3, err
    finally:
        lib.d4_gemm_force_config(-1)


print('v2 configs:', [lib.d4_profile_class_name(n1 + c).decode() for c in range(n2)])
tot1 = tot2 = 0.
for M, N, K, flags, batch, name in shapes:
    fl = 2.0 * M * N * K * batch
    t1 = [bench(M, N, K, flags, batch, c, False)[0] for c in range(n1)]
    r2 = [bench(M, N, K, flags, batch, 100 + c, True) for c in range(n2)]
    b1 = min(t for t in t1 if t is not None)
    v2 = [(t, e) for t, e in r2 if t is not None]
    b2 = min(t for t, _ in v2)
    tot1 += b1; tot2 += min(b1, b2)
    print(f'{name:8s} M{M:6d} N{N:5d} K{K:5d} b{batch} f{flags}: v1 best {b1:7.1f} us {fl / b1 / 1e6:6.1f} TF | v2 best {b2:7.1f} us {fl / b2 / 1e6:6.1f} TF | v2 per cfg: '
          + ' '.join('   --  ' if t is None else f'{t:7.1f}' for t, _ in r2) + ' | max rel err ' + ' '.join('--' if e is None else f'{e:.1e}' for _, e in r2))
print(f'sum of best: v1 {tot1:.0f} us, with v2 where faster {tot2:.0f} us')
