"""Is the SiLU-GLU input projection (FF1, N = 2752, K = 512) on the split-operand kernel paying for a nearly empty second round of workgroups?
Sweeps the row count M (tiles of 128 x 128 = ceil(M / 128) x 22; 512 workgroup slots at two blocks per CU) per tile configuration."""
import ctypes as C, sys
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
RMS, SWIGLU = 1, 4
N, K, flags = 2752, 512, RMS | SWIGLU
names = ['64x64', '128x64', '64x128', '128x128', '128x128/8', '32x64']


def timeit(run, reps=30):
    for _ in range(3):
        run()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator(device='cuda').manual_seed(1)
W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
plane = (N * K + 7) // 8 * 8
W3 = torch.empty(3 * plane, dtype=torch.bfloat16, device='cuda')
_lib.check(lib.d4_split_bf16x3(_lib.ptr(W), _lib.ptr(W3), N * K, plane, s))
b = torch.randn(N, device='cuda', generator=g)
for M in (1792, 2304, 2816, 2944, 3072, 3328, 3584, 3840, 4096, 4608, 5120, 5888, 6144, 7168):
    A = torch.randn(M, K, device='cuda', generator=g)
    Cs = torch.empty(M, N // 2, device='cuda')
    row = []
    for c in (2, 3, 4):
        run = lambda: lib.d4_gemm_split(_lib.ptr(A), K, _lib.ptr(W3), plane, K, _lib.ptr(Cs), N // 2, _lib.ptr(b), None, N, M, N, K, flags, 1.1920929e-07, c, s)
        if run() != 0:
            row.append('   --  '); continue
        t = timeit(run)
        row.append(f'{names[c]} {t:6.1f} us {2. * M * N * K / t / 1e6:6.1f} TF')
    tiles = -(-M // 128) * 22
    print(f'M {M:5d} ({tiles:4d} tiles of 128x128 = {tiles / 512:.2f} rounds): ' + ' | '.join(row), flush=True)
