"""rocprofv3 target (round 6): the headline rollout's per-frame pool with its two phases as separate kernels (d4_frame_fused_set(2): pool_mix_kernel +
frame_pool_tail_kernel) or fused (1: frame_pool_kernel), 4 frames at B = 256.    python tools/frame_pool_phases.py <mode>"""
import sys; sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import DynamicsWorldModel, _lib
from dreamer4_amd.synthetic import randomize_weights
lib = _lib.load()
lib.d4_frame_fused_set(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4), terminal_bias=-10.).cuda()
g = torch.Generator(device='cuda').manual_seed(1234)
m.generate(4, batch_size=256, return_for_policy_optimization=True, generator=g)
torch.cuda.synchronize()
