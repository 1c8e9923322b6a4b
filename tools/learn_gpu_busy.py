"""Is the actor/critic step (learn_from_experience + clip + AdamW on both heads, headline size: 256 trajectories x 16 frames) bound by the GPU or by the host?
Wall time per step against the summed device time of its kernels (torch profiler).    python tools/learn_gpu_busy.py"""
import sys, time; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from torch.profiler import profile, ProfilerActivity
from dreamer4_amd import DreamTrainer, DynamicsWorldModel
from dreamer4_amd.synthetic import randomize_weights
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4), terminal_bias=-10.).cuda()
g = torch.Generator(device='cuda').manual_seed(1234)
e = m.generate(16, batch_size=256, return_for_policy_optimization=True, generator=g)
tr = DreamTrainer(m, batch_size=256, generate_timesteps=15, objective='ppo')
for _ in range(3):
    tr.learn(e)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n):
    tr.learn(e)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / n
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        tr.learn(e)
    torch.cuda.synchronize()
dev = 0.; rows = []
for ev in prof.key_averages():
    t = getattr(ev, 'self_device_time_total', None) or getattr(ev, 'self_cuda_time_total', 0)
    if t > 0:
        dev += t; rows.append((t / 3, ev.count / 3, ev.key))
rows.sort(reverse=True)
print(f'actor/critic step: wall {1e3 * wall:.2f} ms; device time of its kernels {dev / 3 / 1e3:.2f} ms per step ({len(rows)} kinds, {sum(r[1] for r in rows):.0f} launches per step)')
for t, n_, k in rows[:16]:
    print(f'  {t / 1e3:7.3f} ms  x{n_:5.1f}  {k[:110]}')
