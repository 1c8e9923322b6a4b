import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dreamer4_amd import _lib
lib = _lib.load()
import ctypes as C
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in [(64, 64, 512), (128, 128, 96), (256, 128, 1376)]:
    g = torch.Generator(device='cuda').manual_seed(1)
    A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    plane = (N * K + 7) // 8 * 8
    W2 = torch.zeros(2 * plane, dtype=torch.float16, device='cuda'); inv = torch.zeros(N, device='cuda')
    _lib.check(lib.d4_split_f16x2(_lib.ptr(W), _lib.ptr(W2), N, K, K, plane, _lib.ptr(inv), s))
    torch.cuda.synchronize()
    print('shape', M, N, K, 'inv scale log2 range', torch.log2(inv).min().item(), torch.log2(inv).max().item(), 'W row max * s range',
          ((W.abs().amax(1)) / inv).min().item(), ((W.abs().amax(1)) / inv).max().item())
    ref = A.double() @ W.double().t()
    ex = torch.clamp(14 - torch.floor(torch.log2(A.abs().amax(dim=1))), max=126).to(torch.int32)
    for cfg in range(7):
        for use_ex in (False, True):
            o = torch.full((M, N), float('nan'), device='cuda')
            rc = lib.d4_gemm_split2(_lib.ptr(A), K, _lib.ptr(W2), plane, K, _lib.ptr(inv), _lib.ptr(o), N, None, None, N, M, N, K, 0, 0., cfg, _lib.ptr(ex) if use_ex else None, s)
            torch.cuda.synchronize()
            bad = ~torch.isfinite(o)
            err = (o.double() - ref).abs()
            err[bad] = 0
            print(f'  cfg {cfg} a_exp given {use_ex}: rc {rc} nonfinite {int(bad.sum())} rows with nonfinite {bad.any(1).nonzero().flatten().tolist()[:8]} cols {bad.any(0).nonzero().flatten().tolist()[:8]} max err {err.max().item():.2e}')
