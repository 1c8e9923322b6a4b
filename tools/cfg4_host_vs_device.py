"""cfg 4 env step at B = 1: is the step bound by the host (Python + launch enqueue) or by the device?  Times the same 50-call loop
(a) as bench.py does, (b) host-only — wall time until the last call RETURNS, before the final synchronize — and (c) per call with a
synchronize after every call (device latency of one call with an idle queue)."""
import sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import DynamicsWorldModel
from dreamer4_amd.synthetic import randomize_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=512, dim_latent=16, num_latent_tokens=4, num_spatial_tokens=4, depth=6, num_discrete_actions=4), terminal_bias=-10.).cuda()
g = torch.Generator(device='cuda').manual_seed(1)
H = 50
acts = torch.randint(0, 4, (B, H, 1), device='cuda', generator=g)
for mode in ('warm', 'free', 'free', 'sync_each'):
    lat = torch.zeros(B, 0, 4, 16, device='cuda'); rew = torch.zeros(B, 0, device='cuda'); tc = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(H):
        kw = dict(prompt_latents=lat, prompt_discrete_actions=acts[:, :t], prompt_rewards=rew) if t > 0 else {}
        e, tc = m.generate(t + 1, batch_size=B, return_rewards_per_frame=True, return_terminals=True, time_cache=tc, return_time_cache=True, generator=g, **kw)
        lat, rew = e.latents, e.rewards
        if mode == 'sync_each':
            torch.cuda.synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'B={B} {mode}: host loop {1e3 * (t1 - t0) / H:.3f} ms/step, with final sync {1e3 * (t2 - t0) / H:.3f} ms/step')
