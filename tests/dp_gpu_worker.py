"""Worker for tests/test_gpu_dp.py: `world` ranks share cuda:0 over gloo (RCCL refuses two ranks on one device, and
the GPU box has a single MI355X).  Each rank rolls out its shard of a fixed global batch and takes one DreamTrainer
step with global statistics; rank 0 also runs the same global batch alone and compares."""
import os
import sys

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))

from dreamer4_amd import DreamTrainer, parallel      # noqa: E402
from util import first_trajectories, make_noise, oracle_config, small_model  # noqa: E402


def run(model, noise, B, T, group):
    tr = DreamTrainer(model, batch_size=B, generate_timesteps=T - 1, process_group=group, stats='global')
    if B == 0:
        # an EMPTY trajectory shard (6 trajectories over 8 ranks): nothing to roll out, but the rank takes part in every collective of the step with a
        # zero contribution and must end up with the same weights as the others.  (An Experience of zero trajectories: one rolled out, none kept.)
        dreams = first_trajectories(model.generate(T, batch_size=1, return_rewards_per_frame=True, return_agent_actions=True,
                                                   return_log_probs_and_values=True, noise=make_noise(oracle_config(model), T, 1, 7)), 0)
    else:
        dreams = model.generate(T, batch_size=B, return_rewards_per_frame=True, return_agent_actions=True,
                                return_log_probs_and_values=True, noise=noise)
    losses = tr.learn(dreams)
    grads = torch.cat([model._groups['policy']['grad'].cpu(), model._groups['value']['grad'].cpu()])     # after the all-reduce
    return losses.cpu(), grads, torch.cat([p.detach().flatten().cpu() for p in model.policy_head_parameters() if p.numel()]), \
        torch.cat([p.detach().flatten().cpu() for p in model.value_head_parameters()])


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    backend = os.environ.get('D4_DP_BACKEND', 'gloo')          # 'nccl' = RCCL: one device per rank (tests/test_gpu_dp.py, needs >= 2 GPUs)
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    else:
        torch.cuda.set_device(0)
    parallel.init_from_env(backend)
    rank, world = parallel.rank(), parallel.world_size()
    headline = os.environ.get('D4_DP_SIZE', 'small') == 'headline'
    Bg, T = (256, 16) if headline else (6, 4)

    def build():
        if not headline:
            return small_model().cuda()
        # BASELINE config 3's per-rank model: config 2's architecture, the global batch sharded by trajectory (here 2 x 128 of 256)
        from dreamer4_amd import DynamicsWorldModel
        from dreamer4_amd.synthetic import randomize_weights
        torch.manual_seed(0)
        return randomize_weights(DynamicsWorldModel(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4), seed=0, terminal_bias=-3.).cuda()
    small_model_ = build
    m = small_model_()
    cfg = oracle_config(m)
    nz = make_noise(cfg, T, Bg, 55)
    lo, hi = parallel.shard_range(Bg)
    local = {k: v[:, lo:hi].contiguous() for k, v in nz.items()}
    losses, grads, pol, val = run(m, local, hi - lo, T, None)
    if rank == 0:
        ref_model = small_model_()
        # single process over the whole global batch (no process group => world 1 semantics)
        dist_backup = parallel.world_size
        parallel.world_size = lambda group=None: 1
        l_ref, g_ref, p_ref, v_ref = run(ref_model, nz, Bg, T, None)
        parallel.world_size = dist_backup
        assert torch.allclose(losses, l_ref, atol=2e-5), (losses, l_ref)
        # gradients: equal up to the summation order of the two shards
        assert torch.allclose(grads, g_ref, rtol=1e-3, atol=(2e-5 if headline else 1e-6) * float(g_ref.abs().max())), (grads - g_ref).abs().max()
        assert (grads.double() - g_ref.double()).norm() <= 1e-4 * g_ref.double().norm()
        # weights after clip + AdamW: the first Adam step moves every element by ~lr * g / (|g| + eps), so elements whose
        # gradient is at round-off level may differ by a fraction of lr = 3e-4
        assert torch.allclose(pol, p_ref, atol=1e-4) and torch.allclose(val, v_ref, atol=1e-4)
        assert (pol - p_ref).abs().mean() < 1e-6 and (val - v_ref).abs().mean() < 1e-6
    # every rank must hold identical weights after the step
    chk = torch.stack([pol.double().sum(), val.double().sum()])
    if backend == 'nccl':
        chk = chk.cuda()
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    assert all(torch.equal(g, gathered[0]) for g in gathered), gathered
    parallel.barrier()
    os.write(1, f'DPGPU_OK_{rank}|'.encode())


if __name__ == '__main__':
    main()
