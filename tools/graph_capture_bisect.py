"""Which trunk operator breaks torch.cuda.graph capture?  Each case runs in its own process (a failed instantiate can crash it)."""
import subprocess, sys
CASES = ['model_fwd', 'model']
if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True, timeout=280)
        tail = [l for l in (r.stdout + r.stderr).splitlines() if 'Warning' not in l and l.strip()][-2:]
        print(f'{c:16s} rc={r.returncode} {" | ".join(tail)[-260:]}')
    sys.exit(0)
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import DynamicsWorldModel, trunk_ops
from dreamer4_amd.synthetic import randomize_weights
case = sys.argv[1]
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4)).cuda()
W = dict(m.named_parameters()); W.update({k: v for k, v in m.named_buffers() if k.endswith('inv_freq')})
F_, S, D = 64, 15, 512
x = torch.randn(F_, S, D, device='cuda', requires_grad=True)
import os
B, T = int(os.environ.get('BB', 4)), 16
lat = torch.randn(B, T, 32, 32, device='cuda').clamp(-2, 2); acts = torch.randint(0, 4, (B, T, 1), device='cuda')
draws = dict(shortcut_train=False, step_sizes_log2=torch.zeros(B, dtype=torch.long, device='cuda'), signal_levels=torch.randint(0, m.max_steps, (B, T), device='cuda'),
             noise=torch.randn(B, T, 32, 32, device='cuda'))
params = list(m.parameters())


def run():
    for p in params:
        p.grad = None
    x.grad = None
    pre = 'transformer.layers.0.'
    if case.startswith('ff'):
        y = trunk_ops._ff(W, 'transformer.layers.0.1.', x)
    elif case.startswith('space'):
        a = trunk_ops._attn_w(W, 'transformer.layers.0.0.')
        y = trunk_ops.space_attention(x, *a[:7], softclamp_value=50., num_special=1, belief=True) if False else None
    if case in ('ff_fwd', 'ff'):
        if case == 'ff':
            y.sum().backward()
        return y
    if case in ('space_fwd', 'space', 'time', 'cross'):
        is_time = [False] * 6
        if case == 'time':
            is_time = [True] * 6
        tok = x.reshape(4, 16, S, D)
        with torch.set_grad_enabled(case != 'space_fwd'):
            y = trunk_ops.transformer({k: v for k, v in W.items()}, tok, is_time=is_time, softclamp_value=50.) if case != 'cross' else trunk_ops._lq_pool(W, 'latents_to_spatial_tokens.', lat.requires_grad_())
            if case != 'space_fwd':
                y.sum().backward()
        return y
    if case == 'rmsnorm_linear':
        y = trunk_ops._norm_linear(x, W['to_latent_pred.0.weight'], W['to_latent_pred.2.weight']) if False else torch.ops.d4hip.linear(torch.ops.d4hip.rmsnorm(x, W['to_latent_pred.0.weight'], 1e-6), W['to_latent_pred.2.weight'], None, None, 0, 0.)
        y.sum().backward()
        return y
    if case == 'model_fwd':
        with torch.no_grad():
            return m(latents=lat, discrete_actions=acts, draws=draws)
    loss = m(latents=lat, discrete_actions=acts, draws=draws)
    loss.backward()
    return loss


s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        run()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = run()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print('captured + replayed ok', float(out.detach().float().sum()))
