import sys; sys.path.insert(0, '/root/repo')
import ctypes as C, torch
from dreamer4_amd import _lib
lib = _lib.load(); s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def t(f, reps=10):
    for _ in range(3): f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K) in [(8192, 8192, 4096), (3584, 2752, 512)]:
    for mode in ('random', 'zeros', 'ones'):
        g = torch.Generator(device='cuda').manual_seed(1)
        if mode == 'random':
            A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
        elif mode == 'zeros':
            A = torch.zeros(M, K, device='cuda'); W = torch.zeros(N, K, device='cuda')
        else:
            A = torch.ones(M, K, device='cuda'); W = torch.ones(N, K, device='cuda')
        Cs = torch.empty(M, N, device='cuda'); Cn = torch.empty(M, N, device='cuda')
        plane = (N * K + 7) // 8 * 8
        W3 = torch.empty(3 * plane, dtype=torch.bfloat16, device='cuda')
        _lib.check(lib.d4_split_bf16x3(_lib.ptr(W), _lib.ptr(W3), N * K, plane, s))
        x3 = t(lambda: lib.d4_gemm_split(_lib.ptr(A), K, _lib.ptr(W3), plane, K, _lib.ptr(Cs), N, None, None, 0, M, N, K, 0, 1e-6, 4, s))
        nat = t(lambda: lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(Cn), N, None, None, 0, M, N, K, 0, 1e-6, s))
        print(f'M{M} N{N} K{K} {mode:7s}: split (128x128/8) {x3:8.1f} us | native {nat:8.1f} us')
