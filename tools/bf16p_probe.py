"""Probe (round 6): the phased 256 x 256 bf16 GEMM (csrc/gemm_bf16p.hip, configuration 6 of d4_gemm_bf16a) against the other large-tile forms
(7: 256 x 192, 5: 256 x 128) on the config-5 shapes at B = 1024 / B = 128 and on two cubes, random operands.
    python tools/bf16p_probe.py [reps]            interleaved rounds, median and min per configuration"""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ctypes as C
import statistics
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
RMS, SWIGLU = 1, 4
shapes = [(14336, 5504, 1024, RMS | SWIGLU, 'ff1 B=1024'), (14336, 5504, 1024, 0, 'ff1 plain B=1024'), (14336, 1024, 2752, 0, 'ff2 B=1024'),
          (14336, 1552, 1024, RMS, 'proj B=1024'), (14336, 1024, 512, 0, 'out B=1024'), (1792, 5504, 1024, RMS | SWIGLU, 'ff1 B=128'),
          (1792, 1552, 1024, RMS, 'proj B=128'), (4096, 4096, 4096, 0, 'cube 4096'), (8192, 8192, 8192, 0, 'cube 8192'), (8192, 8192, 1024, 0, '8192^2 x 1024')]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
inner = 5
cfgs = (5, 6, 7)


def timed(run):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner * 1e3


for M, N, K, flags, name in shapes:
    g = torch.Generator(device='cuda').manual_seed(1)
    Ab = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16)
    Wb = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).to(torch.bfloat16)
    Nout = N // 2 if flags & SWIGLU else N
    out = torch.empty(M, Nout, device='cuda'); outb = torch.empty(M, Nout, device='cuda', dtype=torch.bfloat16)
    calls = {}
    for c in cfgs:
        # the engine's form of the SiLU-GLU projection writes only the bf16 image (C = null)
        cptr = None if flags & SWIGLU else _lib.ptr(out)
        def call(c=c, cptr=cptr):
            return lib.d4_gemm_bf16a(_lib.ptr(Ab), K, _lib.ptr(Wb), K, cptr, Nout, _lib.ptr(outb), None, None, 0, M, N, K, flags, 1e-6, c, s)
        if call() == 0:
            calls[c] = call
    for c in calls:
        for _ in range(2):
            calls[c]()
    torch.cuda.synchronize()
    ts = {c: [] for c in calls}
    for _ in range(rounds):
        for c in calls:
            ts[c].append(timed(calls[c]))
    fl = 2.0 * M * N * K
    print(f'{name:18s} M{M:6d} N{N:5d} K{K:5d} flags {flags}: ' + ' | '.join(
        f'cfg{c} med {statistics.median(t):7.1f} min {min(t):7.1f} us = {fl / statistics.median(t) / 1e6:6.0f} TF/s' for c, t in ts.items()), flush=True)
