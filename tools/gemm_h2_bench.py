"""GPU microbenchmark: the split-operand fp32 GEMM (gemm_h2.hip: two fp16 planes per operand under exact row scales, three fp16 MFMA
products, fp32 accumulate) against the f32-input MFMA families on the engine's cfg-2 shapes, per tile configuration, with the error of
both against float64.      python tools/gemm_h2_bench.py [reps]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from dreamer4_amd import _lib

lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
RMS, SILU, SWIGLU = 1, 2, 4
shapes = [(3584, 1552, 512, RMS, 'proj'), (3584, 2064, 512, RMS, 'proj0'), (3584, 512, 512, 0, 'out'), (3584, 2752, 512, RMS | SWIGLU, 'ff1'),
          (3584, 512, 1376, 0, 'ff2'), (3584, 256, 512, RMS, 'poolq'), (10752, 256, 512, RMS, 'poolk3'), (25088, 256, 512, RMS, 'poolk7'),
          (39424, 256, 512, RMS, 'poolk11'), (3584, 512, 256, 0, 'poolout'), (3840, 2752, 512, RMS | SWIGLU, 'ff1c'),
          (1024, 512, 512, 0, 'c_out'), (1024, 2752, 512, RMS | SWIGLU, 'c_ff1'), (13312, 256, 512, RMS, 'c_poolk'),
          (8192, 1024, 32, RMS, 'lkv'), (8192, 32, 512, 0, 'lout'), (4096, 2048, 2048, 0, 'headL'), (8192, 8192, 4096, 0, 'big')]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
NX = 7
names = ['64x64', '128x64', '64x128', '128x128', '128x128/8', '32x64', '64x128/o3']


def timeit(run):
    for _ in range(3):
        run()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print('split configs:', names)
tot_n = tot_s = 0.
for M, N, K, flags, name in shapes:
    g = torch.Generator(device='cuda').manual_seed(1)
    A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    Nout = N // 2 if flags & SWIGLU else N
    Cn = torch.full((M, Nout), float('nan'), device='cuda'); Cs = torch.full((M, Nout), float('nan'), device='cuda')
    b = torch.randn(N, device='cuda', generator=g)
    R = torch.randn(M, N, device='cuda', generator=g) if not (flags & SWIGLU) else None
    plane = (N * K + 7) // 8 * 8
    W3 = torch.empty(2 * plane, dtype=torch.float16, device='cuda')
    inv = torch.empty(N, device='cuda')
    _lib.check(lib.d4_split_f16x2(_lib.ptr(W), _lib.ptr(W3), N, K, K, plane, _lib.ptr(inv), s))
    native = lambda: lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(Cn), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, 1.1920929e-07, s)
    _lib.check(native())
    ref = None
    if M * N * K <= 4096 * 2048 * 2048:
        Ad, Wd = A.double(), W.double()
        X = Ad * torch.rsqrt(Ad.pow(2).mean(-1, keepdim=True) + 1.1920929e-07) if flags & RMS else Ad
        ref = X @ Wd.t() + b.double()
        if flags & SWIGLU:
            r = ref.reshape(M, N // 64, 2, 32)
            ref = (r[:, :, 0] * torch.nn.functional.silu(r[:, :, 1])).reshape(M, N // 2)
        if R is not None:
            ref = ref + R.double()
    tn = timeit(native)
    ts, first = [], None
    same = True
    aexp = torch.zeros(M, dtype=torch.int32, device='cuda')
    texp = timeit(lambda: lib.d4_row_scale_exp(_lib.ptr(A), K, M, K, _lib.ptr(aexp), s))
    t_pro = {}
    for c in list(range(NX)) + [100 + c for c in range(NX)]:
        Cs.fill_(float('nan'))
        cc = c % 100
        run = lambda: lib.d4_gemm_split2(_lib.ptr(A), K, _lib.ptr(W3), plane, K, _lib.ptr(inv), _lib.ptr(Cs), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, 1.1920929e-07, cc, _lib.ptr(aexp) if c < 100 else None, s)
        if c >= 100:          # the same tile with the kernel finding the row exponents itself (a pass over the A panel per column tile)
            if run() == 0:
                torch.cuda.synchronize(); t_pro[cc] = timeit(run)
            continue
        if run() != 0:
            ts.append(None); continue
        torch.cuda.synchronize()
        if first is None: first = Cs.clone()
        else: same = same and torch.equal(first, Cs)
        ts.append(timeit(run))
    bs = min(t for t in ts if t is not None)
    fl = 2.0 * M * N * K
    tot_n += tn; tot_s += bs
    errs = ''
    if ref is not None:
        sc = ref.abs().mean().item()
        en = (Cn.double() - ref); es = (first.double() - ref)
        errs = f' | rms err / mean|ref|: native {en.pow(2).mean().sqrt().item() / sc:.2e} split {es.pow(2).mean().sqrt().item() / sc:.2e}; max: {en.abs().max().item() / sc:.2e} {es.abs().max().item() / sc:.2e}'
    b3 = min(t_pro.values()) if t_pro else float('nan')
    print(f'{name:8s} M{M:6d} N{N:5d} K{K:5d} f{flags}: native {tn:7.1f} us {fl / tn / 1e6:6.1f} TF | row exponents given: best {bs:7.1f} us {fl / bs / 1e6:6.1f} TF (x{tn / bs:.2f}) | per cfg: '
          + ' '.join('   --  ' if t is None else f'{t:7.1f}' for t in ts) + f' | exponents found in the kernel: best {b3:7.1f} | row-exponent kernel {texp:5.1f} us | same bits {same}' + errs, flush=True)
print(f'sum: native {tot_n:.0f} us, split best {tot_s:.0f} us')
