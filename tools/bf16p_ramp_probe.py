"""Probe (round 6): the phased 256 x 256 bf16 kernel's time as a + b K on config 5's SiLU-GLU input projection shape (14336 x 5504, folded RMSNorm, bf16 image only) and on the
plain epilogue: which part of a launch is the k-loop and which is per-tile prologue / epilogue.    python tools/bf16p_ramp_probe.py"""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ctypes as C
import statistics
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N = 14336, 5504


def timed(run, inner=5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner * 1e3


for flags, name in ((5, 'RMS + SiLU-GLU, bf16 image only'), (0, 'plain, fp32 + bf16 image'), (0x10000, 'plain, bf16 image only')):
    only_b = flags != 0
    fl = flags & 0xFFFF
    ts = {}
    for K in (256, 512, 1024, 2048, 4096):
        g = torch.Generator(device='cuda').manual_seed(1)
        Ab = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16)
        Wb = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).to(torch.bfloat16)
        Nout = N // 2 if fl & 4 else N
        out = torch.empty(M, Nout, device='cuda'); outb = torch.empty(M, Nout, device='cuda', dtype=torch.bfloat16)
        call = lambda: lib.d4_gemm_bf16a(_lib.ptr(Ab), K, _lib.ptr(Wb), K, None if only_b else _lib.ptr(out), Nout, _lib.ptr(outb), None, None, 0, M, N, K, fl, 1e-6, 6, s)
        assert call() == 0
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        ts[K] = statistics.median(timed(call) for _ in range(7))
    b = (ts[4096] - ts[1024]) / (4096 - 1024)
    a = ts[1024] - b * 1024
    rounds = -(-(M // 256) * -(-N // 256) // 256)
    print(f'{name:34s}: ' + '  '.join(f'K={K}: {t:6.1f} us' for K, t in ts.items()) + f'   | t = {a:5.1f} + {b * 1024:5.1f} per 1024 of K  ->  k-loop {2.0 * M * N / b / 1e6:6.0f} TF/s, '
          f'fixed part {a:.1f} us = {100 * a / ts[1024]:.0f} % of the K = 1024 launch ({rounds} rounds of tiles: {a / rounds:.1f} us per round)', flush=True)
