"""GPU: wall time of the cfg-2 rollout (B=256, 16 frames) per trunk GEMM arithmetic: the default fp32 path and the opt-in fp16x2 mode (gemm_h2.hip).
Median of 5 after two warm passes; same weights, same draws; also the largest difference of the two rollouts."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreamer4_amd import DynamicsWorldModel
from dreamer4_amd.synthetic import randomize_weights
kw = dict(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4)
out = {}
for mode in sys.argv[1:] or ['fp32', 'fp32_fp16x2', 'fp32', 'fp32_fp16x2']:
    torch.manual_seed(0)
    m = randomize_weights(DynamicsWorldModel(**kw, matmul_dtype=mode), terminal_bias=-10.).cuda()
    g = torch.Generator(device='cuda').manual_seed(1234)
    for _ in range(2): m.generate(16, batch_size=256, return_for_policy_optimization=True, generator=g)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e = m.generate(16, batch_size=256, return_for_policy_optimization=True, generator=g)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    g = torch.Generator(device='cuda').manual_seed(99)
    e = m.generate(16, batch_size=256, return_for_policy_optimization=True, generator=g)
    if mode in out:
        pass
    elif out:
        ref = next(iter(out.values()))
        print(f'   vs {next(iter(out))}: latents max |diff| {(e.latents - ref.latents).abs().max().item():.2e}, values {(e.values - ref.values).abs().max().item():.2e}, '
              f'actions equal on {int((e.actions.discrete == ref.actions.discrete).flatten(1).all(1).sum())} / 256 trajectories')
    out.setdefault(mode, e)
    print(f'{mode:12s} rollout {1e3 * sorted(ts)[2]:.2f} ms (min {1e3 * min(ts):.2f})', flush=True)
    del m
    torch.cuda.empty_cache()
