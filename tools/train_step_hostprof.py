"""Where the HOST time of a dynamics training step goes: cProfile over 5 eager steps (cfg 2 architecture, B=16 x T=16)."""
import cProfile, pstats, sys
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import DynamicsWorldModel
from dreamer4_amd.synthetic import randomize_weights
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4)).cuda()
g = torch.Generator(device='cuda').manual_seed(1)
lat = torch.randn(16, 16, 32, 32, device='cuda', generator=g).clamp(-2, 2)
acts = torch.randint(0, 4, (16, 16, 1), device='cuda', generator=g)
params = list(m.parameters())


def step():
    for p in params:
        p.grad = None
    m(latents=lat, discrete_actions=acts, generator=g, prob_shortcut_train=0.).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
