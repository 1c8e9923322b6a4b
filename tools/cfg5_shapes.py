"""Per-shape table of the bf16 trunk GEMMs of one cfg-5 rollout (dim 1024, depth 12, B = 128, 6 frames): D4_GEMM_LOG=1 python tools/cfg5_shapes.py
(table on stderr; per-launch HIP events force the eager launch mechanism, the kernels are the default path's)."""
import ctypes as C, sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import DynamicsWorldModel, _lib
from dreamer4_amd.synthetic import randomize_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=1024, dim_latent=32, num_latent_tokens=64, depth=12, num_continuous_actions=6, matmul_dtype='bf16'), terminal_bias=-10.).cuda()
g = torch.Generator(device='cuda').manual_seed(1234)
gk = dict(return_for_policy_optimization=True, num_steps=4, generator=g)
lib = _lib.load()
m.generate(2, batch_size=B, **gk)
torch.cuda.synchronize(); t0 = time.perf_counter()
m.generate(6, batch_size=B, **gk)
torch.cuda.synchronize()
print(f'default path: {1e3 * (time.perf_counter() - t0) / 6:.2f} ms per frame')
lib.d4_profile_bf16_enable(1)
m.generate(6, batch_size=B, **gk)
torch.cuda.synchronize()
lib.d4_profile_bf16_enable(0)
ms, fl, cnt = C.c_double(), C.c_double(), C.c_int64()
_lib.check(lib.d4_profile_bf16_read(C.byref(ms), C.byref(fl), C.byref(cnt)))
print(f'bf16 GEMM launches {cnt.value}, {ms.value:.1f} ms over 6 frames ({ms.value / 6:.2f} ms per frame), {fl.value / ms.value / 1e9:.1f} TF/s')
