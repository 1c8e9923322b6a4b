"""Weight-gradient GEMM (csrc/gemm_tn.hip) against the transposed-operand form of d4_gemm on the shapes of a cfg-2 training step, with a
sweep of the tile / slice choices.   python tools/gemm_tn_bench.py"""
import ctypes as C, sys
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import _lib
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
GEMM_TRANS_A, GEMM_TRANS_B = 8, 16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


shapes = [(2736, 512, 3840), (512, 1368, 3840), (1568, 512, 3840), (512, 512, 3840), (512, 256, 3840), (260, 512, 3840), (512, 512, 49920), (512, 32, 8192), (32, 512, 8192)]
part = torch.empty(16 << 20, device='cuda')
for M, N, K in shapes:
    A = torch.randn(K, M, device='cuda'); B = torch.randn(K, N, device='cuda')
    Cn = torch.empty(M, N, device='cuda'); Co = torch.empty(M, N, device='cuda')
    ref = (A.double().T @ B.double())
    fl = 2. * M * N * K
    res = []
    for tn in (1, 3, 4, 5):
        for S in (0, 2, 4, 6, 8):
            if S * M * N > part.numel():
                continue
            f = lambda: _lib.check(lib.d4_gemm_tn(_lib.ptr(A), M, _lib.ptr(B), N, _lib.ptr(Cn), N, M, N, K, _lib.ptr(part), part.numel(), tn if S else 0, S, st))
            us = timeit(f)
            err = ((Cn.double() - ref).abs().max() / ref.abs().max()).item()
            res.append((us, tn, S, err))
    best = min(res)
    rule = [r for r in res if r[2] == 0][0]
    old = timeit(lambda: _lib.check(lib.d4_gemm(_lib.ptr(A), M, _lib.ptr(B), N, _lib.ptr(Co), N, None, None, 0, M, N, K, GEMM_TRANS_A | GEMM_TRANS_B, 0., st)))
    print(f'M {M:5d} N {N:5d} K {K:6d}: rule {rule[0]:7.1f} us ({fl / rule[0] / 1e6:6.1f} TF/s, err {rule[3]:.1e}) | best tn={best[1]} S={best[2]} {best[0]:7.1f} us ({fl / best[0] / 1e6:6.1f} TF/s) '
          f'| d4_gemm transposed (no split) {old:7.1f} us | ' + ' '.join(f'{tn}/{S}:{us:.0f}' for us, tn, S, _ in res))
