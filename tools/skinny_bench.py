"""GPU microbenchmark: the few-row GEMM on BASELINE config 4's shapes (B = 1: 11 token rows; B = 16: 176 rows), replayed from a
hipGraph of 48 launches that rotate over enough weight copies to miss the L2s (the engine streams ~90 MB of weights per evaluation).
    python tools/skinny_bench.py            -> us per launch (includes the ~1.8 us kernel boundary of a graph)"""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ctypes as C
import torch
from dreamer4_amd import _lib

lib = _lib.load()
RMS, SILU, SWIGLU = 1, 2, 4
EPS = 1.1920929e-07
shapes = [('proj', 1552, 512, RMS), ('out', 512, 512, 0), ('ff1', 2752, 512, RMS | SWIGLU), ('ff2', 512, 1376, 0), ('poolq', 256, 512, RMS),
          ('poolout', 512, 256, 0), ('tinyK', 512, 16, 0)]
Ms = [int(a) for a in sys.argv[1:]] or [11, 143, 176]
NL = 48


def bench(M, N, K, flags):
    g = torch.Generator(device='cuda').manual_seed(1)
    ncopy = max(2, min(NL, (96 << 20) // (N * K * 4)))
    W = torch.randn(ncopy, N, K, device='cuda', generator=g)
    A = torch.randn(M, K, device='cuda', generator=g)
    Nout = N // 2 if flags & SWIGLU else N
    Cc = torch.zeros(M, Nout, device='cuda')
    b = torch.randn(N, device='cuda', generator=g)
    R = torch.randn(M, N, device='cuda', generator=g) if not (flags & SWIGLU) else None
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        def launch(i):
            s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            rc = lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W[i % ncopy]), K, _lib.ptr(Cc), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, EPS, s)
            assert rc == 0
        launch(0); torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for i in range(NL):
                launch(i)
    torch.cuda.synchronize()
    for _ in range(3):
        graph.replay()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (reps * NL)


for M in Ms:
    print(f'M={M}: ' + '  '.join(f'{nm}({N}x{K}) {bench(M, N, K, fl):.2f}us' for nm, N, K, fl in shapes), flush=True)
