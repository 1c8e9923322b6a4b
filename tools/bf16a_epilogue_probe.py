"""How much of a short-K bf16 GEMM is its epilogue traffic?  (M = 14336, N = 1024, K = 512 / 256: the residual-stream projections of cfg 5 at B = 1024)"""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ctypes as C
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(run, reps=30):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, N, K in ((14336, 1024, 512), (14336, 1024, 256), (1792, 1024, 512)):
    g = torch.Generator(device='cuda').manual_seed(1)
    Ab = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16); Wb = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device='cuda'); outb = torch.empty(M, N, device='cuda', dtype=torch.bfloat16); R = torch.randn(M, N, device='cuda', generator=g)
    for name, c, cb, r in (('C + Cb + R', out, outb, R), ('C + Cb', out, outb, None), ('C only', out, None, None), ('Cb only', None, outb, None), ('C + R', out, None, R)):
        byt = M * N * ((4 if c is not None else 0) + (2 if cb is not None else 0) + (4 if r is not None else 0)) + (M + N) * K * 2
        for cfg in (2, 1, 0):
            t = timeit(lambda: lib.d4_gemm_bf16a_batched(_lib.ptr(Ab), K, _lib.ptr(Wb), K, _lib.ptr(c), N, _lib.ptr(cb), None, _lib.ptr(r), N, M, N, K, 0, 1e-6, 1, 0, 0, 0, cfg, s))
            print(f'M{M:6d} N{N:5d} K{K:4d} {name:11s} cfg {cfg}: {t:7.1f} us  {byt / t / 1e6:6.2f} TB/s of operand + epilogue bytes, {2.0 * M * N * K / t / 1e6:6.0f} TF/s')
