"""GPU: continuous (Beta) actions and the second head-MLP recipe on the HIP engine, against the fixtures frozen from the
reference (through the Readout / BetaDist / normed-MLP stand-ins, oracle/shim) and against the oracle on fresh seeds.
Sampled continuous actions are floating point: tolerance 1e-5 (the accept / reject decisions of the gamma sampler are made
with a margin >= 2e-3 in every fixture, so they agree exactly); discrete indices, lens and terminals stay bit-exact."""
import numpy as np
import pytest
import torch

from dreamer4_amd import Actions, Experience
from oracle import restate
from util import golden_model, golden_noise, golden_oracle, load_golden, make_noise, oracle_config, oracle_weights, randomize_weights, t

pytestmark = pytest.mark.gpu


def close(a, b, atol=2e-4, rtol=1e-4):
    a = a.detach().float().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).float()
    b = b.detach().float().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).float()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, atol=atol, rtol=rtol), f'max abs diff {(a - b).abs().max().item():.3e} (scale {b.abs().max().item():.3e})'


def check_rollout(e, g, prefix):
    assert e.latents.shape[1] == g[prefix + 'latents'].shape[1]
    close(e.latents, g[prefix + 'latents']); close(e.agent_embed, g[prefix + 'agent_embed'])
    close(e.rewards, g[prefix + 'rewards']); close(e.values, g[prefix + 'values'])
    assert np.array_equal(e.lens.cpu().numpy(), g[prefix + 'lens']) and np.array_equal(e.terminals.cpu().numpy(), g[prefix + 'terminals'])
    if prefix + 'actions' in g:
        assert np.array_equal(e.actions.discrete.cpu().numpy(), g[prefix + 'actions'])
        close(e.log_probs.discrete, g[prefix + 'log_probs']); close(e.old_action_unembeds.discrete, g[prefix + 'unembeds'])
    if prefix + 'actions_cont' in g:
        close(e.actions.continuous, g[prefix + 'actions_cont'], atol=1e-5)
        close(e.log_probs.continuous, g[prefix + 'log_probs_cont'], atol=2e-4)
        close(e.old_action_unembeds.continuous, g[prefix + 'cont_params'])


def check_learn(m, exp, g, objectives, min_grads):
    P = dict(m.named_parameters())
    for obj in objectives:
        m.zero_grad()
        pl, vl = m.learn_from_experience(exp, objective=obj)
        close(pl, g[f'{obj}_policy_loss'], atol=1e-5); close(vl, g[f'{obj}_value_loss'], atol=1e-5)
        pl.backward(retain_graph=True); vl.backward()
        n = 0
        for k, v in g.items():
            name = k.split('/', 1)[1] if '/' in k else None
            if k.startswith(f'{obj}_grad/'):
                ref = t(v)
                close(P[name].grad, ref, atol=2e-6 + 1e-4 * ref.abs().max().item(), rtol=1e-3); n += 1
            elif k.startswith(f'{obj}_gnorm/'):
                close(P[name].grad.norm(), v, atol=1e-6, rtol=1e-3); n += 1
            elif k.startswith(f'{obj}_gsample/'):
                ref = t(v)
                close(P[name].grad.flatten()[::97], ref, atol=2e-6 + 1e-4 * ref.abs().max().item(), rtol=1e-3)
        assert n >= min_grads


def test_linear_layernorm_head_recipe_vs_reference_fixture():
    g = load_golden('postln.npz')
    m = golden_model('weights_postln.npz').cuda()
    assert m.head_mlp_recipe == 'post_layer'
    e = m.generate(5, batch_size=3, return_for_policy_optimization=True, noise=golden_noise(g, 'cached_'))
    check_rollout(e, g, 'cached_')
    check_learn(m, e, g, ('ppo', 'pmpo'), 12)


def test_continuous_actions_mixed_model_vs_reference_fixture():
    g = load_golden('continuous.npz')
    m = golden_model('weights_continuous.npz').cuda()
    assert m.num_continuous_actions == 3
    e = m.generate(5, batch_size=3, return_for_policy_optimization=True, noise=golden_noise(g, 'cached_'))
    check_rollout(e, g, 'cached_')
    # learn from the FIXTURE's experience (so the comparison does not inherit the rollout's rounding)
    exp = Experience(latents=t(g['cached_latents']), agent_embed=t(g['cached_agent_embed']), rewards=t(g['cached_rewards']), values=t(g['cached_values']),
                     log_probs=Actions(t(g['cached_log_probs']), t(g['cached_log_probs_cont'])), actions=Actions(t(g['cached_actions']), t(g['cached_actions_cont'])),
                     lens=t(g['cached_lens']), terminals=t(g['cached_terminals']), is_truncated=~t(g['cached_terminals']),
                     old_action_unembeds=Actions(t(g['cached_unembeds']), t(g['cached_cont_params'])), step_size=16)
    check_learn(m, exp, g, ('ppo', 'spo', 'pmpo'), 10)
    assert m.action_embedder.continuous_action_unembed.grad.abs().max() > 0


def test_beta_head_exp_link_vs_reference_fixture():
    """The Beta head's link as a descriptor (continuous_beta_param='exp_p1'): rollout, losses and head gradients of the HIP engine against
    the fixture frozen from the reference with the stand-in switched to the exp link."""
    g = load_golden('beta_exp.npz')
    m = golden_model('weights_beta_exp.npz').cuda()
    assert m.continuous_beta_param == 'exp_p1' and m.num_continuous_actions == 3
    e = m.generate(5, batch_size=3, return_for_policy_optimization=True, continuous_temperature=0.8, noise=golden_noise(g, 'cached_'))
    check_rollout(e, g, 'cached_')
    exp = Experience(latents=t(g['cached_latents']), agent_embed=t(g['cached_agent_embed']), rewards=t(g['cached_rewards']), values=t(g['cached_values']),
                     log_probs=Actions(t(g['cached_log_probs']), t(g['cached_log_probs_cont'])), actions=Actions(t(g['cached_actions']), t(g['cached_actions_cont'])),
                     lens=t(g['cached_lens']), terminals=t(g['cached_terminals']), is_truncated=~t(g['cached_terminals']),
                     old_action_unembeds=Actions(t(g['cached_unembeds']), t(g['cached_cont_params'])), step_size=16)
    check_learn(m, exp, g, ('ppo', 'pmpo'), 10)
    assert m.action_embedder.continuous_action_unembed.grad.abs().max() > 0


def test_continuous_only_model_vs_reference_fixture():
    g = load_golden('continuous.npz')
    m = golden_model('weights_contonly.npz').cuda()
    e = m.generate(4, batch_size=2, return_for_policy_optimization=True, continuous_temperature=0.7, noise=golden_noise(g, 'only_'))
    check_rollout(e, g, 'only_')
    assert e.actions.discrete is None and e.log_probs.discrete is None
    pl, vl = m.learn_from_experience(e, objective='ppo')
    close(pl, g['only_ppo_policy_loss'], atol=1e-5); close(vl, g['only_ppo_value_loss'], atol=1e-5)
    # env-wrapper pattern (dreamer4/env.py:445-483) with continuous prompts and the carried time cache
    nz = golden_noise(g, 'env_')
    lat = torch.zeros(2, 0, 3, 4); act = torch.zeros(2, 0, 2); tc = None
    for i in range(3):
        sub = {k: v[i:i + 1] for k, v in nz.items()}
        kw = dict(prompt_latents=lat, prompt_continuous_actions=act) if i > 0 else {}
        e, tc = m.generate(i + 1, batch_size=2, return_rewards_per_frame=True, return_agent_actions=True, return_log_probs_and_values=True,
                           time_cache=tc, return_time_cache=True, noise=sub, **kw)
        close(e.latents, g[f'env{i}_latents']); close(e.actions.continuous, g[f'env{i}_actions_cont'], atol=1e-5); close(e.values, g[f'env{i}_values'])
        lat, act = e.latents.cpu(), e.actions.continuous.cpu()


@pytest.mark.parametrize('kw', [dict(num_continuous_actions=4), dict(num_continuous_actions=2, num_discrete_actions=0, head_mlp_recipe='post_layer'),
                                dict(num_continuous_actions=1, num_discrete_actions=(3, 2), depth=3, time_block_every=1),
                                dict(num_continuous_actions=3, continuous_beta_param='exp_p1')])
def test_continuous_generate_and_learn_vs_oracle_on_fresh_models(kw):
    from dreamer4_amd import DynamicsWorldModel
    base = dict(dim=64, dim_latent=8, num_latent_tokens=6, depth=4, time_block_every=2, attn_heads=2, num_discrete_actions=4, num_tasks=0)
    base.update(kw)
    torch.manual_seed(1)
    m = randomize_weights(DynamicsWorldModel(**base), seed=3)
    with torch.no_grad():
        m.action_embedder.continuous_action_unembed.mul_(30.)
    cfg, W = oracle_config(m), oracle_weights(m)
    B, T = 3, 4
    for seed in range(50, 90):            # a seed whose gamma accept / reject decisions carry a margin (well-posed exactness)
        nz = make_noise(cfg, T, B, seed)
        ref = restate.generate(cfg, W, T, batch_size=B, noise=nz, return_terminals=False)
        if restate.beta_accept_margin(ref['old_cont_params'], nz['beta'][:ref['old_cont_params'].shape[1]].transpose(0, 1), 1., cfg.continuous_beta_param) >= 2e-3:
            break
    m = m.cuda()
    e = m.generate(T, batch_size=B, return_rewards_per_frame=True, return_agent_actions=True, return_log_probs_and_values=True, noise=nz)
    close(e.latents, ref['latents']); close(e.values, ref['values'])
    close(e.actions.continuous, ref['actions_cont'], atol=1e-5); close(e.log_probs.continuous, ref['log_probs_cont'], atol=2e-4)
    if cfg.num_discrete_actions:
        assert torch.equal(e.actions.discrete.cpu(), ref['actions'])
    heads = ('policy_head', 'value_head', 'action_embedder.discrete_action_unembed', 'action_embedder.continuous_action_unembed')
    P = dict(m.named_parameters())
    for obj in ('ppo', 'pmpo'):
        Wg = {k: (v.clone().requires_grad_() if k.startswith(heads) and v.numel() > 0 else v) for k, v in W.items()}
        pl_o, vl_o = restate.learn_losses(cfg, Wg, ref, obj)
        pl_o.backward(); vl_o.backward()
        m.zero_grad()
        pl, vl = m.learn_from_experience(e, objective=obj)
        close(pl, pl_o, atol=2e-5); close(vl, vl_o, atol=2e-5)
        pl.backward(retain_graph=True); vl.backward()
        for k, v in Wg.items():
            if v.requires_grad and v.grad is not None:
                close(P[k].grad, v.grad, atol=5e-6 + 2e-4 * v.grad.abs().max().item(), rtol=2e-3)


def test_learn_without_stored_agent_embeddings_vs_reference_fixture():
    """generate(store_agent_embed=False): learn_from_experience recomputes the agent embeddings with one parallel forward over the
    stored latents (dreamer4.py:6045-6070) — here through d4_wm_forward."""
    g = load_golden('postln.npz')
    m = golden_model('weights_postln.npz').cuda()
    exp = Experience(latents=t(g['noembed_latents']), agent_embed=None, rewards=t(g['noembed_rewards']), values=t(g['noembed_values']),
                     log_probs=Actions(t(g['noembed_log_probs']), None), actions=Actions(t(g['noembed_actions']), None), lens=t(g['noembed_lens']),
                     terminals=t(g['noembed_terminals']), is_truncated=~t(g['noembed_terminals']),
                     old_action_unembeds=Actions(t(g['noembed_unembeds']), None), step_size=16)
    pl, vl = m.learn_from_experience(exp, objective='ppo')
    close(pl, g['noembed_ppo_policy_loss'], atol=1e-5); close(vl, g['noembed_ppo_value_loss'], atol=1e-5)
    pl.backward(retain_graph=True); vl.backward()
    ref = t(g['noembed_ppo_grad_unembed'])
    close(m.action_embedder.discrete_action_unembed.grad, ref, atol=2e-6 + 1e-4 * ref.abs().max().item(), rtol=1e-3)


def test_symexp_two_hot_encoder_vs_reference_fixture():
    g = load_golden('symexp.npz')
    m = golden_model('weights_symexp.npz').cuda()
    assert m.reward_encoder_type == 'symexp_two_hot' and 'value_encoder.bin_values' in m.state_dict()
    e = m.generate(5, batch_size=3, return_for_policy_optimization=True, noise=golden_noise(g, 'cached_'))
    check_rollout(e, g, 'cached_')
    check_learn(m, e, g, ('ppo',), 8)


@pytest.mark.parametrize('link', ['softplus_p1', 'exp_p1'])
def test_beta_link_descriptor_in_the_training_forward_continuous_loss(link):
    """The mirror's continuous behaviour-cloning term (dreamer4.py:7566-7597) honours `continuous_beta_param` like the kernels and the
    oracle do; an unknown link is refused at construction."""
    import dataclasses
    from dreamer4_amd import trunk_ops
    from oracle import restate
    cfg, W = golden_oracle('weights_beta_exp.npz')
    cfg = dataclasses.replace(cfg, continuous_beta_param=link)
    g = torch.Generator().manual_seed(7)
    b, t_ = 2, 5
    agent = torch.randn(b, t_, cfg.dim, generator=g)
    lat = torch.randn(b, t_, cfg.num_latent_tokens, cfg.dim_latent, generator=g)
    ca = torch.rand(b, t_, cfg.num_continuous_actions, generator=g)
    Wd = {k: v.cuda() for k, v in W.items()}
    ref = restate.dynamics_agent_losses(cfg, W, agent, lat, cont_actions=ca)
    out = trunk_ops.dynamics_agent_losses(
        Wd, agent.cuda(), lat.cuda(), multi_token_pred_len=cfg.multi_token_pred_len, num_discrete_actions=tuple(cfg.num_discrete_actions), reward_range=cfg.reward_range,
        reward_num_bins=cfg.reward_num_bins, policy_head_mlp_depth=cfg.policy_head_mlp_depth, terminal_mlp_depth=cfg.terminal_mlp_depth,
        head_mlp_recipe=cfg.head_mlp_recipe, continuous_beta_param=link, continuous_actions=ca.cuda())
    assert torch.allclose(out['continuous_actions'].cpu(), ref['continuous_actions'], atol=2e-6, rtol=2e-5)
    other = restate.dynamics_agent_losses(dataclasses.replace(cfg, continuous_beta_param='exp_p1' if link == 'softplus_p1' else 'softplus_p1'), W, agent, lat, cont_actions=ca)
    assert (other['continuous_actions'] - ref['continuous_actions']).abs().max() > 1e-3
    with pytest.raises(ValueError, match='continuous_beta_param'):
        from dreamer4_amd import DynamicsWorldModel
        DynamicsWorldModel(dim=32, dim_latent=8, num_latent_tokens=4, depth=2, num_continuous_actions=2, continuous_beta_param='sigmoid')


def test_learn_at_config5_size_vs_oracle():
    """BASELINE config 5's actor/critic step (bench.py `cfg5_bf16.actor_critic_step_ms`): learn_from_experience(ppo) on the continuous (Beta) head at dim 1024,
    B = 128 trajectories x 16 frames = 2048 learner rows, against the oracle on ONE Experience (the GPU's own fp32 rollout of a depth-2 trunk of that width —
    the learner only sees agent embeddings, so the trunk's depth is irrelevant to it): both losses and every gradient of policy_head.*, value_head.* and
    continuous_action_unembed, with the tolerance taken from the oracle's own sensitivity to a 2e-6 rounding of its inputs (as the config-2 test)."""
    from dreamer4_amd import Actions, DynamicsWorldModel
    torch.set_num_threads(min(16, torch.get_num_threads()))
    torch.manual_seed(0)
    m = randomize_weights(DynamicsWorldModel(dim=1024, dim_latent=32, num_latent_tokens=64, depth=2, time_block_every=2, num_continuous_actions=6), terminal_bias=-10.)
    cfg, W = oracle_config(m), oracle_weights(m)
    B, T = 128, 16
    m = m.cuda()
    e = m.generate(T, batch_size=B, return_for_policy_optimization=True, noise=make_noise(cfg, T, B, 1234))
    assert e.agent_embed.shape == (B, T, 1024) and e.actions.continuous.shape[-1] == 6
    cpu = lambda x: x.detach().cpu()
    ref = dict(latents=cpu(e.latents), agent_embed=cpu(e.agent_embed), rewards=cpu(e.rewards), values=cpu(e.values), log_probs_cont=cpu(e.log_probs.continuous),
               actions_cont=cpu(e.actions.continuous), lens=cpu(e.lens), terminals=cpu(e.terminals), is_truncated=cpu(e.is_truncated),
               old_cont_params=cpu(e.old_action_unembeds.continuous), step_size=e.step_size)
    heads = ('policy_head', 'value_head', 'action_embedder.continuous_action_unembed')
    grads = lambda: {k: (v.clone().requires_grad_() if k.startswith(heads) and v.numel() > 0 else v) for k, v in W.items()}
    Wg = grads()
    pl_o, vl_o = restate.learn_losses(cfg, Wg, ref, 'ppo')
    pl_o.backward(); vl_o.backward()
    gen = torch.Generator().manual_seed(7)
    jig = lambda x, r: x * (1 + r * (2 * torch.rand(x.shape, generator=gen) - 1))
    ref2 = dict(ref, values=jig(ref['values'], 2e-6), rewards=jig(ref['rewards'], 2e-6), agent_embed=jig(ref['agent_embed'], 2e-6),
                log_probs_cont=ref['log_probs_cont'] + 2e-6 * ref['log_probs_cont'].abs().max() * (2 * torch.rand(ref['log_probs_cont'].shape, generator=gen) - 1))
    Wp = grads()
    pl_p, vl_p = restate.learn_losses(cfg, Wp, ref2, 'ppo')
    pl_p.backward(); vl_p.backward()
    pl, vl = m.learn_from_experience(e, objective='ppo')
    close(pl, pl_o, atol=2e-5 + 20 * abs(pl_p.item() - pl_o.item())); close(vl, vl_o, atol=2e-5 + 20 * abs(vl_p.item() - vl_o.item()))
    pl.backward(retain_graph=True); vl.backward()
    P = dict(m.named_parameters())
    n = 0
    for k, v in Wg.items():
        if v.requires_grad and v.grad is not None:
            go, gg = v.grad.double(), P[k].grad.detach().cpu().double()
            sens = (Wp[k].grad.double() - go).norm().item()
            assert (gg - go).norm().item() <= 20 * sens + 1e-5 * go.norm().item() + 1e-12, (k, (gg - go).norm().item(), sens, go.norm().item())
            n += 1
    assert n >= 10
