import sys, time; sys.path.insert(0, '/root/repo')
import torch
from dreamer4_amd import DynamicsWorldModel
from dreamer4_amd.synthetic import randomize_weights
torch.manual_seed(0)
m = DynamicsWorldModel(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4)
randomize_weights(m, terminal_bias=-10.)
m = m.cuda()
g = torch.Generator(device='cuda').manual_seed(1234)
gen = m.generate(4, batch_size=256, return_for_policy_optimization=True, generator=g)   # 4 frames = 20 evaluations
torch.cuda.synchronize()
