"""rocprofv3 target: the actor-critic step of the headline alone — one rollout (B=256, H=15), then 12 x (learn_from_experience(ppo) +
clip + AdamW on both heads) on the same dreams."""
import sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import DynamicsWorldModel, DreamTrainer
from dreamer4_amd.synthetic import randomize_weights
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4), terminal_bias=-10.).cuda()
tr = DreamTrainer(m, batch_size=256, generate_timesteps=15)
dreams = tr.generate()
for _ in range(3):
    tr.learn(dreams)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(12):
    tr.learn(dreams)
torch.cuda.synchronize()
print(f'{1e3 * (time.perf_counter() - t0) / 12:.2f} ms per actor-critic step')
