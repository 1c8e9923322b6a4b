"""Probe (round 6): tile configurations of the bf16-activation GEMM on the wide key projection's shapes (folded RMSNorm, bf16 image only, leading dimension of the
output = depth x 256) at 1792 and 14336 token rows per slab.    python tools/bf16a_widekeys_probe.py"""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ctypes as C
import statistics
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
K, LD, depth = 1024, 12 * 256, 12


def timed(run, inner=5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner * 1e3


for rows in (1792, 14336):
    for p in range(depth - 1):
        M = (3 if p == 0 else 2) * rows; N = (depth - p) * 256
        g = torch.Generator(device='cuda').manual_seed(1)
        Ab = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16)
        Wb = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).to(torch.bfloat16)
        outb = torch.empty(M, LD, device='cuda', dtype=torch.bfloat16)
        calls = {}
        for c in range(-1, 8):
            def call(c=c):
                return lib.d4_gemm_bf16a(_lib.ptr(Ab), K, _lib.ptr(Wb), K, None, LD, _lib.ptr(outb), None, None, 0, M, N, K, 1, 1e-6, c, s)
            if call() == 0:
                calls[c] = call
        for c in calls:
            calls[c]()
        torch.cuda.synchronize()
        ts = {c: [] for c in calls}
        for _ in range(4):
            for c in calls:
                ts[c].append(timed(calls[c]))
        med = {c: statistics.median(t) for c, t in ts.items()}
        best = min((c for c in med if c >= 0), key=lambda c: med[c])
        print(f'M{M:6d} N{N:5d}: ' + ' '.join(f'c{c} {t:6.1f}' for c, t in med.items()) + f' | rule {med[-1]:6.1f} best c{best} {med[best]:6.1f} ({100 * (med[-1] / med[best] - 1):+.0f} %)', flush=True)
