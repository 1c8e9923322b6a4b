"""GPU microbenchmark + bit check: the A-stationary / W-streaming split-operand GEMM (gemm_x3w.hip) against the tiled forms of the same arithmetic
(gemm_x3.hip configurations 0-5, 7 = the persistent form the engine uses) and the f32-input MFMA kernels, on the cfg-2 shapes.   python tools/gemm_x3w_bench.py [reps]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from dreamer4_amd import _lib

lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
RMS, SILU, SWIGLU = 1, 2, 4
shapes = [(3584, 2752, 512, RMS | SWIGLU, 'ff1'), (3840, 2752, 512, RMS | SWIGLU, 'ff1c'), (3584, 1552, 512, RMS, 'proj'), (3584, 2064, 512, RMS, 'proj0'),
          (3584, 512, 1376, 0, 'ff2'), (3584, 512, 512, 0, 'out'), (3584, 256, 512, RMS, 'poolq'), (10752, 256, 512, RMS, 'poolk3'), (25088, 256, 512, RMS, 'poolk7'),
          (39424, 256, 512, RMS, 'poolk11'), (46592, 256, 512, RMS, 'poolk13'), (1024, 2752, 512, RMS | SWIGLU, 'c_ff1'), (4096, 2048, 2048, 0, 'headL'),
          (1000, 300, 96, RMS, 'ragged'), (130, 129, 2048, 0, 'ragged2'), (8192, 8192, 4096, 0, 'big')]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


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
    b = torch.randn(N, device='cuda', generator=g)
    R = torch.randn(M, N, device='cuda', generator=g) if not (flags & SWIGLU) else None
    plane = (N * K + 7) // 8 * 8
    W3 = torch.empty(3 * plane, dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.d4_split_bf16x3(_lib.ptr(W), _lib.ptr(W3), N * K, plane, s))
    Wt = torch.empty(lib.d4_split_bf16x3_tiled_elems(N, K), dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.d4_split_bf16x3_tiled(_lib.ptr(W), _lib.ptr(Wt), N, K, K, s))
    Cn = torch.full((M, Nout), float('nan'), device='cuda'); Cx = torch.full((M, Nout), float('nan'), device='cuda'); Cw = torch.full((M, Nout), float('nan'), device='cuda')
    native = lambda: lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(Cn), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, 1.1920929e-07, s)
    tiled = lambda c: (lambda: lib.d4_gemm_split(_lib.ptr(A), K, _lib.ptr(W3), plane, K, _lib.ptr(Cx), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, 1.1920929e-07, c, s))
    stream_w = lambda: lib.d4_gemm_splitw(_lib.ptr(A), K, _lib.ptr(Wt), _lib.ptr(Cw), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, 1.1920929e-07, s)
    _lib.check(native()); _lib.check(stream_w())
    tt = {}
    for c in (0, 1, 2, 3, 4, 5, 7):
        if tiled(c)() == 0:
            torch.cuda.synchronize(); tt[c] = timeit(tiled(c))
    c0 = min(tt, key=tt.get)
    _lib.check(tiled(2 if flags & SWIGLU else 0)()); torch.cuda.synchronize()
    same = torch.equal(Cx, Cw)
    tn, tw = timeit(native), timeit(stream_w)
    fl = 2.0 * M * N * K
    print(f'{name:8s} M{M:6d} N{N:5d} K{K:5d} f{flags}: f32-input {tn:7.1f} us {fl / tn / 1e6:6.1f} TF | tiled bf16x3 best {tt[c0]:7.1f} us (cfg {c0}) {fl / tt[c0] / 1e6:6.1f} TF | '
          f'W-streaming {tw:7.1f} us {fl / tw / 1e6:6.1f} TF (x{tt[c0] / tw:.2f} tiled, x{tn / tw:.2f} f32-input) | bit-identical to the tiled form: {same}'
          + ('' if same else f' max |diff| {(Cx - Cw).abs().max().item():.2e} nan {int(torch.isnan(Cw).sum())}'), flush=True)
