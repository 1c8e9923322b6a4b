"""Time one dynamics training step (flow + shortcut losses, forward + backward through the HIP trunk blocks) at BASELINE config 2's
architecture: dim 512, depth 6, 8 x 64 heads, 32 x 32 latents; B x T frames of 15 tokens.   python tools/train_step_time.py [B] [T]"""
import sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import DynamicsWorldModel
from dreamer4_amd.synthetic import randomize_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4)).cuda()
g = torch.Generator(device='cuda').manual_seed(1)
lat = torch.randn(B, T, 32, 32, device='cuda', generator=g).clamp(-2, 2)
acts = torch.randint(0, 4, (B, T, 1), device='cuda', generator=g)
params = [p for p in m.parameters()]
for shortcut in (0., 1.):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            for p in params:
                p.grad = None
            loss = m(latents=lat, discrete_actions=acts, generator=g, prob_shortcut_train=shortcut)
            loss.backward()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    rows = B * T * 15
    print(f'B={B} T={T} ({rows} token rows) shortcut={int(shortcut)}: {1e3 * dt:.1f} ms per training step (forward + backward{" + 2 no-grad target forwards" if shortcut else ""}), '
          f'{B * T / dt:.0f} frames/s')
if len(sys.argv) > 3:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        loss = m(latents=lat, discrete_actions=acts, generator=g, prob_shortcut_train=0.)
        loss.backward()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=25, max_name_column_width=70))
