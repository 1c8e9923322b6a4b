"""rocprofv3 target: BASELINE config 4's env-step pattern at B = 1 (dim 512, depth 6, 4 x 16 latents), 30 steps after a warm pass."""
import sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import DynamicsWorldModel
from dreamer4_amd.synthetic import randomize_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=512, dim_latent=16, num_latent_tokens=4, num_spatial_tokens=4, depth=6, num_discrete_actions=4), terminal_bias=-10.).cuda()
g = torch.Generator(device='cuda').manual_seed(1)
H = 30
acts = torch.randint(0, 4, (B, H, 1), device='cuda', generator=g)
for rep in range(2):
    lat = torch.zeros(B, 0, 4, 16, device='cuda'); rew = torch.zeros(B, 0, device='cuda'); tc = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(H):
        kw = dict(prompt_latents=lat, prompt_discrete_actions=acts[:, :t], prompt_rewards=rew) if t > 0 else {}
        e, tc = m.generate(t + 1, batch_size=B, return_rewards_per_frame=True, return_terminals=True, time_cache=tc, return_time_cache=True, generator=g, **kw)
        lat, rew = e.latents, e.rewards
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'B={B}: {1e3 * dt / H:.3f} ms per env step')
