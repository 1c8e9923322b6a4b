"""GPU: wall time of the cfg-2 rollout (B=256, 16 frames), median of 5 after a warm pass."""
import sys, time; sys.path.insert(0, '/root/repo')
import torch
from dreamer4_amd import DynamicsWorldModel
from dreamer4_amd.synthetic import randomize_weights
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4), terminal_bias=-10.).cuda()
g = torch.Generator(device='cuda').manual_seed(1234)
for _ in range(2): m.generate(16, batch_size=256, return_for_policy_optimization=True, generator=g)
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.generate(16, batch_size=256, return_for_policy_optimization=True, generator=g)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f'rollout {1e3 * sorted(ts)[2]:.2f} ms (min {1e3 * min(ts):.2f})')
