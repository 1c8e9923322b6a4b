"""Is gemm_bf16a's k-loop bound by where its operands come from?  Same tile (128x128, cfg 0), growing footprints: time per k-tile step."""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ctypes as C
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(run, reps=30):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for cfg, bm, bn in ((0, 128, 128), (5, 256, 128), (6, 256, 256)):
    for M, N, K in ((2048, 2048, 1024), (2048, 2048, 4096), (2048, 2048, 16384), (4096, 4096, 1024), (4096, 4096, 4096), (1792, 5504, 1024), (1792, 5504, 4096), (8192, 8192, 1024), (1024, 1024, 16384)):
        g = torch.Generator(device='cuda').manual_seed(1)
        Ab = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16); Wb = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).to(torch.bfloat16)
        out = torch.empty(M, N, device='cuda')
        t = timeit(lambda: lib.d4_gemm_bf16a(_lib.ptr(Ab), K, _lib.ptr(Wb), K, _lib.ptr(out), N, None, None, None, 0, M, N, K, 0, 1e-6, cfg, s))
        tiles = -(-M // bm) * -(-N // bn)
        slots = 256
        rounds = -(-tiles // slots)
        print(f'cfg {cfg} ({bm}x{bn}) M{M:5d} N{N:5d} K{K:6d}: {t:8.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF/s  tiles {tiles:5d} rounds {rounds} -> {1e3 * t / (rounds * K / 64):7.1f} ns per k-tile step; operands {(M + N) * K * 2 / 1e6:6.1f} MB')
