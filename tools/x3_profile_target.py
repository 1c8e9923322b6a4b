"""GPU: one shape of the split-operand GEMM, a fixed tile configuration, 20 launches (rocprofv3 --pmc target).
    python tools/x3_profile_target.py M N K flags config"""
import sys; sys.path.insert(0, '/root/repo')
import ctypes as C
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K, flags, cfg = (int(x) for x in sys.argv[1:6])
g = torch.Generator(device='cuda').manual_seed(1)
A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
Nout = N // 2 if flags & 4 else N
Cs = torch.empty(M, Nout, device='cuda'); b = torch.randn(N, device='cuda', generator=g)
plane = (N * K + 7) // 8 * 8
W3 = torch.empty(3 * plane, dtype=torch.bfloat16, device='cuda')
_lib.check(lib.d4_split_bf16x3(_lib.ptr(W), _lib.ptr(W3), N * K, plane, s))
for _ in range(20):
    _lib.check(lib.d4_gemm_split(_lib.ptr(A), K, _lib.ptr(W3), plane, K, _lib.ptr(Cs), Nout, _lib.ptr(b), None, 0, M, N, K, flags, 1.1920929e-07, cfg, s))
torch.cuda.synchronize()
