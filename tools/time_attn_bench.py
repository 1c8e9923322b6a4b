"""GPU microbenchmark of the time-attention decode kernel at the cfg-2 shape (B=256, S=14, H=8) for growing cache lengths."""
import sys, os; sys.path.insert(0, '/root/repo')
import torch, time
from dreamer4_amd import DynamicsWorldModel
from dreamer4_amd.synthetic import randomize_weights
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4), terminal_bias=-10.).cuda()
g = torch.Generator(device='cuda').manual_seed(1)
for T in (4, 16, 50):
    m.generate(T, batch_size=256, return_for_policy_optimization=True, generator=g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.generate(T, batch_size=256, return_for_policy_optimization=True, generator=g)
    torch.cuda.synchronize()
    print(f'T={T}: {1e3 * (time.perf_counter() - t0) / T:.2f} ms per frame', 'legacy' if os.environ.get('D4_TIME_ATTN_LEGACY') else 'new')
