"""rocprofv3 target: BASELINE config 5's rollout (dim 1024, depth 12, 64 x 32 latents, 6 continuous actions, B = 128 or argv[2], bf16 trunk GEMMs), 4 frames after a warm pass."""
import sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import DynamicsWorldModel
from dreamer4_amd.synthetic import randomize_weights

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=1024, dim_latent=32, num_latent_tokens=64, depth=12, num_continuous_actions=6, matmul_dtype=dtype), terminal_bias=-10.).cuda()
g = torch.Generator(device='cuda').manual_seed(1234)
gk = dict(return_for_policy_optimization=True, num_steps=4, generator=g)
m.generate(2, batch_size=B, **gk)
torch.cuda.synchronize(); t0 = time.perf_counter()
m.generate(4, batch_size=B, **gk)
torch.cuda.synchronize()
print(f'{dtype} B={B}: {1e3 * (time.perf_counter() - t0) / 4:.2f} ms per frame')
