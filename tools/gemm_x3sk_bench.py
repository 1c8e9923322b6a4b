"""GPU microbenchmark: the persistent / k-cut form of the split-operand fp32 GEMM (gemm_x3sk.hip, d4_gemm_split config 6) against the
plain 128 x 128 form (config 4) and the f32-input MFMA families, on the cfg-2 shapes.    python tools/gemm_x3sk_bench.py [reps]"""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ctypes as C
import torch
from dreamer4_amd import _lib

lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
RMS, SILU, SWIGLU = 1, 2, 4
shapes = [(3584, 2752, 512, RMS | SWIGLU, 'ff1'), (3840, 2752, 512, RMS | SWIGLU, 'ff1c'), (3584, 1552, 512, RMS, 'proj'), (3840, 1552, 512, RMS, 'projc'),
          (3584, 2064, 512, RMS, 'proj0'), (3584, 512, 1376, 0, 'ff2'), (3840, 512, 1376, 0, 'ff2c'), (3584, 512, 512, 0, 'out'),
          (1024, 2752, 512, RMS | SWIGLU, 'c_ff1'), (1024, 512, 1376, 0, 'c_ff2'), (39424, 256, 512, RMS, 'poolk11'), (4096, 2048, 2048, 0, 'headL'),
          (1792, 1024, 1024, RMS, 'cfg5 out'), (1792, 5504, 1024, RMS | SWIGLU, 'cfg5 ff1'), (1792, 1024, 2752, 0, 'cfg5 ff2')]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
EPS = 1.1920929e-07


def timeit(run):
    for _ in range(3):
        run()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M, N, K, flags, name in shapes:
    g = torch.Generator(device='cuda').manual_seed(1)
    A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    Nout = N // 2 if flags & SWIGLU else N
    outs = [torch.full((M, Nout), float('nan'), device='cuda') for _ in range(4)]
    b = torch.randn(N, device='cuda', generator=g)
    R = torch.randn(M, N, device='cuda', generator=g) if not (flags & SWIGLU) else None
    plane = (N * K + 7) // 8 * 8
    W3 = torch.empty(3 * plane, dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.d4_split_bf16x3(_lib.ptr(W), _lib.ptr(W3), N * K, plane, s))
    native = lambda: lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(outs[0]), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, EPS, s)
    split = lambda c, o: lib.d4_gemm_split(_lib.ptr(A), K, _lib.ptr(W3), plane, K, _lib.ptr(o), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, EPS, c, s)
    _lib.check(native()); _lib.check(split(4, outs[1])); _lib.check(split(6, outs[2])); _lib.check(split(7, outs[3]))
    torch.cuda.synchronize()
    again = torch.full((M, Nout), float('nan'), device='cuda')
    _lib.check(split(6, again)); torch.cuda.synchronize()
    Ad, Wd = A.double(), W.double()
    X = Ad * torch.rsqrt(Ad.pow(2).mean(-1, keepdim=True) + EPS) if flags & RMS else Ad
    ref = X @ Wd.t() + b.double()
    if flags & SWIGLU:
        r = ref.reshape(M, N // 64, 2, 32)
        ref = (r[:, :, 0] * torch.nn.functional.silu(r[:, :, 1])).reshape(M, N // 2)
    if R is not None:
        ref = ref + R.double()
    sc = ref.abs().mean().item()
    err = [((o.double() - ref).pow(2).mean().sqrt().item() / sc) for o in outs]
    dmax = (outs[2] - outs[1]).abs().max().item() / sc
    same_frac = (outs[2] == outs[1]).float().mean().item()
    half_same = torch.equal(outs[3], outs[1])
    tn, t4, t6, t7 = timeit(native), timeit(lambda: split(4, outs[1])), timeit(lambda: split(6, outs[2])), timeit(lambda: split(7, outs[3]))
    fl = 2.0 * M * N * K
    print(f'{name:9s} M{M:6d} N{N:5d} K{K:5d} f{flags}: native {tn:6.1f} us {fl / tn / 1e6:6.1f} TF | 128x128/8 {t4:6.1f} us {fl / t4 / 1e6:6.1f} TF | half tiles {t7:6.1f} us {fl / t7 / 1e6:6.1f} TF (x{t4 / t7:.2f} plain, bits equal {half_same}) | k-cut {t6:6.1f} us '
          f'{fl / t6 / 1e6:6.1f} TF (x{tn / t6:.2f} native, x{t4 / t6:.2f} plain) | rms err native {err[0]:.2e} plain {err[1]:.2e} persistent {err[2]:.2e} | '
          f'max |persistent - plain| {dmax:.1e}, equal {100 * same_frac:.1f} % | repeat identical {torch.equal(again, outs[2])} finite {bool(torch.isfinite(outs[2]).all())}', flush=True)
