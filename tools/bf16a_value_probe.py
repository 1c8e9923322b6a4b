"""Probe (round 6): tile configurations on the attention pool's per-head value projection as the bf16 engine calls it (batch = 4 heads: A [M][4 x 1024] head slices at
stride 1024, W [4][64][1024], output [M][256] bf16 only, 64 columns per head), 1792 and 14336 rows.    python tools/bf16a_value_probe.py"""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ctypes as C
import statistics
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
K, H = 1024, 4


def timed(run, inner=5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner * 1e3


for M in (1792, 14336):
    g = torch.Generator(device='cuda').manual_seed(1)
    Ab = torch.randn(M, H * K, device='cuda', generator=g).to(torch.bfloat16)
    Wb = (torch.randn(H * 64, K, device='cuda', generator=g) / K ** 0.5).to(torch.bfloat16)
    outb = torch.empty(M, H * 64, device='cuda', dtype=torch.bfloat16)
    calls = {}
    for c in range(-1, 8):
        def call(c=c):
            return lib.d4_gemm_bf16a_batched(_lib.ptr(Ab), H * K, _lib.ptr(Wb), K, None, H * 64, _lib.ptr(outb), None, None, 0, M, 64, K, 0, 1e-6, H, K, 64 * K, 64, c, s)
        if call() == 0:
            calls[c] = call
    for c in calls:
        calls[c]()
    torch.cuda.synchronize()
    ts = {c: [] for c in calls}
    for _ in range(5):
        for c in calls:
            ts[c].append(timed(calls[c]))
    med = {c: statistics.median(t) for c, t in ts.items()}
    print(f'value projection M{M:6d}: ' + ' '.join(f'c{c} {t:6.1f}' for c, t in med.items()) + ' us', flush=True)
