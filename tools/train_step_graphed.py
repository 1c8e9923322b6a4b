"""Experiment: the whole dynamics training step (forward + backward, flow loss) captured once in a HIP graph through torch.cuda.graph
and replayed — how much of the eager step is host time?   python tools/train_step_graphed.py [B] [T]"""
import sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import DynamicsWorldModel
from dreamer4_amd.synthetic import randomize_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4)).cuda()
lat = torch.randn(B, T, 32, 32, device='cuda').clamp(-2, 2)
acts = torch.randint(0, 4, (B, T, 1), device='cuda')
params = list(m.parameters())
draws = dict(shortcut_train=False, step_sizes_log2=torch.zeros(B, dtype=torch.long, device='cuda'), signal_levels=torch.randint(0, m.max_steps, (B, T), device='cuda'),
             noise=torch.randn(B, T, 32, 32, device='cuda'))


def step():
    for p in params:
        p.grad = None
    loss = m(latents=lat, discrete_actions=acts, draws=draws)
    loss.backward()
    return loss


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        loss = step()
    torch.cuda.synchronize()
torch.cuda.current_stream().wait_stream(s)
print(f'eager: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms per step, loss {float(loss.detach()):.6f}')
del loss
g = torch.cuda.CUDAGraph()
for p in params:
    p.grad = None
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print(f'graphed: {1e3 * (time.perf_counter() - t0) / 10:.2f} ms per step, loss {float(static_loss):.6f}, grads finite '
      f'{all(torch.isfinite(p.grad).all().item() for p in params if p.grad is not None)}')
