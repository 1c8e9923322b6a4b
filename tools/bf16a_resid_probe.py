"""Probe (round 6): the bf16-activation GEMM's tile configurations on config 5's OUTPUT projections as the engine calls them — fp32 residual read,
fp32 output + its bf16 image written (10 bytes per output element beside 2 K flops: the K <= 512 ones are bound by that traffic, not by the MFMAs) —
and on the plain projections, at 1792 and 14336 token rows.  Interleaved rounds, median per configuration.    python tools/bf16a_resid_probe.py [rounds]"""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ctypes as C
import statistics
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
RMS = 1
# (M, N, K, flags, residual, fp32 out, name)
shapes = []
for M in (1792, 14336):
    shapes += [(M, 1024, 256, 0, True, True, 'pool out'), (M, 1024, 512, 0, True, True, 'attn out'), (M, 1024, 2752, 0, True, True, 'ff out'),
               (M, 1552, 1024, RMS, False, True, 'qkv proj')]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
inner = 5


def timed(run):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner * 1e3


for M, N, K, flags, resid, f32out, name in shapes:
    g = torch.Generator(device='cuda').manual_seed(1)
    Ab = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16)
    Wb = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).to(torch.bfloat16)
    R = torch.randn(M, N, device='cuda', generator=g) if resid else None
    out = torch.empty(M, N, device='cuda'); outb = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    calls = {}
    for c in range(-1, 8):                 # -1: the shape rule
        def call(c=c):
            return lib.d4_gemm_bf16a(_lib.ptr(Ab), K, _lib.ptr(Wb), K, _lib.ptr(out) if f32out else None, N, _lib.ptr(outb), None, _lib.ptr(R), N, M, N, K, flags, 1e-6, c, s)
        if call() == 0:
            calls[c] = call
    for c in calls:
        calls[c]()
    torch.cuda.synchronize()
    ts = {c: [] for c in calls}
    for _ in range(rounds):
        for c in calls:
            ts[c].append(timed(calls[c]))
    byts = M * N * ((4 if resid else 0) + (4 if f32out else 0) + 2) + 2 * K * (M + N)
    print(f'{name:9s} M{M:6d} N{N:5d} K{K:5d}: ' + ' '.join(f'c{c} {statistics.median(t):6.1f}' for c, t in ts.items()) +
          f' us | bytes {byts / 1e6:.0f} MB = {byts / min(statistics.median(t) for t in ts.values()) / 1e6:.2f} TB/s at the best', flush=True)
