"""rocprofv3 --pmc target: one cfg-5 GEMM shape on gemm_bf16a, a given tile configuration, 10 launches.  python tools/bf16a_pmc_target.py ff1 0"""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ctypes as C
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = dict(ff1=(1792, 5504, 1024, 5), ff2=(1792, 1024, 2752, 0), out=(1792, 1024, 512, 0), proj=(1792, 1552, 1024, 1), cube=(4096, 4096, 4096, 0),
              ff1L=(14336, 5504, 1024, 5), ff2L=(14336, 1024, 2752, 0), cube8=(8192, 8192, 8192, 0))
M, N, K, flags = shapes[sys.argv[1]]; c = int(sys.argv[2])
g = torch.Generator(device='cuda').manual_seed(1)
Ab = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16); Wb = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).to(torch.bfloat16)
Nout = N // 2 if flags & 4 else N
out = torch.empty(M, Nout, device='cuda'); outb = torch.empty(M, Nout, device='cuda', dtype=torch.bfloat16)
for _ in range(10):
    assert lib.d4_gemm_bf16a(_lib.ptr(Ab), K, _lib.ptr(Wb), K, (None if flags & 4 else _lib.ptr(out)), Nout, _lib.ptr(outb), None, None, 0, M, N, K, flags, 1e-6, c, s) == 0
torch.cuda.synchronize()
