"""Probe: gemm_bf16a tile configurations x epilogue flags on the cfg-5 shapes (which part of a launch costs what)."""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ctypes as C
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
RMS, SWIGLU = 1, 4
shapes = [(14336, 5504, 1024, 'ff1L'), (14336, 1024, 2752, 'ff2L'), (14336, 1552, 1024, 'projL'), (114688, 256, 1024, 'poolkL'), (1792, 5504, 1024, 'ff1'), (1792, 1024, 2752, 'ff2'), (1792, 1024, 512, 'out'), (1792, 1552, 1024, 'proj'), (1792, 256, 1024, 'poolq'), (1792, 1024, 256, 'poolout'), (23296, 256, 1024, 'poolk13')]
reps = 30
def timeit(run):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
NC = lib.d4_gemm_bf16a_configs() if hasattr(lib, 'd4_gemm_bf16a_configs') else 6
for M, N, K, name in shapes:
    g = torch.Generator(device='cuda').manual_seed(1)
    Ab = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16); Wb = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device='cuda'); outb = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    for flags in (0, RMS | SWIGLU):
        if (flags & SWIGLU) and N % 64: continue
        Nout = N // 2 if flags & SWIGLU else N
        for cb in (True,):
            ts = []
            for c in range(8):
                call = lambda: lib.d4_gemm_bf16a(_lib.ptr(Ab), K, _lib.ptr(Wb), K, _lib.ptr(out), Nout, _lib.ptr(outb) if cb else None, None, None, 0, M, N, K, flags, 1e-6, c, s)
                ts.append(timeit(call) if call() == 0 else float('nan'))
            print(f'{name:8s} M{M:6d} N{N:5d} K{K:5d} flags {flags} Cb {int(cb)}: ' + ' '.join(f'{t:7.1f}' for t in ts) + f' | best {2.0 * M * N * K / min(t for t in ts if t == t) / 1e6:6.0f} TF/s')
