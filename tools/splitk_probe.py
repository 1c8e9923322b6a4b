import os, sys; sys.path.insert(0, os.getcwd())
import ctypes as C, torch
from dreamer4_amd import _lib
lib = _lib.load(); s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(run, reps=50):
    for _ in range(5): run()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K) in [(256, 2048, 2048), (256, 2048, 512), (256, 255, 2048), (256, 2048, 256)]:
    g = torch.Generator(device='cuda').manual_seed(1)
    A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    Cn = torch.empty(M, N, device='cuda'); b = torch.randn(N, device='cuda', generator=g)
    Np = (N + 3) // 4 * 4
    t1 = timeit(lambda: lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(Cn), N, _lib.ptr(b), None, 0, M, N, K, 2, 0., s))
    out = [f'{M}x{N}x{K}: direct (bias + SiLU in the epilogue) {t1:.1f} us']
    for S in (2, 4, 8):
        if K % (S * 32): continue
        part = torch.empty(S, M, N, device='cuda')
        t = timeit(lambda: lib.d4_gemm_batched(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(part), N, None, None, 0, M, N, K // S, 0, 0., S, K // S, K // S, M * N, s))
        out.append(f'k-split {S}: {t:.1f} us (+ reduce ~5)')
    print(' | '.join(out), flush=True)
