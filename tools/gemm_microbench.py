import sys; sys.path.insert(0, '/root/repo')
import torch, ctypes as C
from dreamer4_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [(16384,512,512,1,'fit1'), (16384,1024,512,1,'fit2'), (32768,1024,512,1,'fit4'), (16384,1024,2048,1,'fit2k'), (3840,1552,512,1,'proj'), (3840,2064,512,1,'proj0'), (3840,2752,512,5,'ff1'), (3840,512,1376,0,'ff2'), (3840,512,512,0,'out'),
          (11520,512,512,1,'poolkv3'), (26880,512,512,1,'poolkv7'), (49920,512,512,1,'poolkv13'), (3840,260,512,1,'poolq'), (3840,512,256,0,'poolout'),
          (3840,1024,512,1,'ckv'), (8192,1024,32,1,'lkv'), (8192,512,512,0,'oproj'), (256,2048,2048,2,'head')]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tot_f = tot_t = 0
for M,N,K,flags,name in shapes:
    A = torch.randn(M,K,device='cuda'); W = torch.randn(N,K,device='cuda'); b = torch.randn(N,device='cuda')
    Nout = N//2 if flags & 4 else N
    Cc = torch.empty(M,Nout,device='cuda')
    def run(): _lib.check(lib.d4_gemm(_lib.ptr(A),K,_lib.ptr(W),K,_lib.ptr(Cc),Nout,_lib.ptr(b),None,0,M,N,K,flags,1e-7,s))
    for _ in range(3): run()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/reps*1e3
    fl = 2.0*M*N*K
    Wt = W.t().contiguous()
    for _ in range(3): torch.matmul(A, W.t())
    e0.record()
    for _ in range(reps): torch.matmul(A, W.t())
    e1.record(); torch.cuda.synchronize()
    us_nt = e0.elapsed_time(e1)/reps*1e3
    for _ in range(3): torch.matmul(A, Wt)
    e0.record()
    for _ in range(reps): torch.matmul(A, Wt)
    e1.record(); torch.cuda.synchronize()
    us_nn = e0.elapsed_time(e1)/reps*1e3
    print(f'{name:10s} M{M:6d} N{N:5d} K{K:5d}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s   | library sgemm (plain, no epilogue) NT {fl/us_nt/1e6:6.1f}  NN {fl/us_nn/1e6:6.1f} TF/s')
    tot_f += fl; tot_t += us
print(f'sum: {tot_t:.0f} us, {tot_f/tot_t/1e6:.1f} TF/s')
