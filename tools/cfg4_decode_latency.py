"""BASELINE config 4 driving pattern (dreamer4/env.py:445-483): Snake 4x4-style action-conditioned world model,
dim=512 depth=6, 4 discrete actions, synthetic latents (4 tokens x 16: one spatial token per latent token), one generated frame per call with the
KV-cached time state carried across calls, horizon 50.  Prints ms per env step and steps/s for B=1 and B=16."""
import sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import DynamicsWorldModel
from dreamer4_amd.synthetic import randomize_weights

torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=512, dim_latent=16, num_latent_tokens=4, num_spatial_tokens=4, depth=6, num_discrete_actions=4),
                      terminal_bias=-10.).cuda()
H = 50
for B in (1, 16):
    g = torch.Generator(device='cuda').manual_seed(1)
    for rep in range(2):
        lat = torch.zeros(B, 0, 4, 16, device='cuda'); act = torch.zeros(B, 0, 1, dtype=torch.long, device='cuda')
        tc = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(H):
            kw = dict(prompt_latents=lat, prompt_discrete_actions=act) if t > 0 else {}
            e, tc = m.generate(t + 1, batch_size=B, return_rewards_per_frame=True, return_agent_actions=True,
                               return_log_probs_and_values=True, time_cache=tc, return_time_cache=True, generator=g, **kw)
            lat, act = e.latents, e.actions.discrete
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'cfg4 B={B}: {1e3 * dt / H:.2f} ms per env step, {B * H / dt:.0f} imagined steps/s (horizon {H})')
