"""Per-shape GEMM table of one cfg-2 rollout (B=256, H=15): D4_GEMM_LOG=1 python tools/rollout_shapes.py  (table on stderr)."""
import ctypes as C, sys
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import DynamicsWorldModel, _lib
from dreamer4_amd.synthetic import randomize_weights
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4), terminal_bias=-10.).cuda()
g = torch.Generator(device='cuda').manual_seed(1234)
lib = _lib.load()
for _ in range(2):
    m.generate(16, batch_size=256, return_for_policy_optimization=True, generator=g)
torch.cuda.synchronize()
ncls = lib.d4_profile_classes()
lib.d4_profile_enable((1 << ncls) - 1)
m.generate(16, batch_size=256, return_for_policy_optimization=True, generator=g)
torch.cuda.synchronize()
lib.d4_profile_enable(0)
ms = (C.c_double * ncls)(); fl = (C.c_double * ncls)(); cnt = (C.c_int64 * ncls)()
_lib.check(lib.d4_profile_read(ms, fl, cnt, ncls))
print(f'GEMM launches {sum(cnt)}, {sum(ms):.1f} ms, {sum(fl) / sum(ms) / 1e9:.1f} TF/s')
