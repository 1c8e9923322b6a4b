"""GPU: learn_from_experience (losses + gradients of both heads) and the native clipped AdamW step against the
reference fixtures and the oracle."""
import numpy as np
import pytest
import torch

from dreamer4_amd import Actions, DreamTrainer, Experience
from oracle import restate
from util import golden_model, golden_noise, golden_oracle, load_golden, make_noise, oracle_config, oracle_weights, small_model, t

pytestmark = pytest.mark.gpu
HEADS = ('policy_head', 'value_head', 'action_embedder.discrete_action_unembed')


def close(a, b, atol=1e-5, rtol=1e-4):
    a = a.detach().float().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).float()
    b = b.detach().float().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).float()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, atol=atol, rtol=rtol), f'max abs diff {(a - b).abs().max().item():.3e} (scale {b.abs().max().item():.3e})'


def fixture_experience(G):
    return Experience(latents=t(G['cached_latents']), agent_embed=t(G['cached_agent_embed']), rewards=t(G['cached_rewards']),
                      values=t(G['cached_values']), log_probs=Actions(t(G['cached_log_probs']), None),
                      actions=Actions(t(G['cached_actions']), None), lens=t(G['cached_lens']), terminals=t(G['cached_terminals']),
                      is_truncated=~t(G['cached_terminals']), old_action_unembeds=Actions(t(G['cached_unembeds']), None), step_size=16)


@pytest.mark.parametrize('objective', ['ppo', 'spo', 'pmpo'])
def test_learn_vs_reference_fixture(objective):
    m = golden_model().cuda()
    G, g = load_golden('generate.npz'), load_golden('learn.npz')
    pl, vl = m.learn_from_experience(fixture_experience(G), objective=objective)
    close(pl, g[f'{objective}_policy_loss'], atol=1e-5); close(vl, g[f'{objective}_value_loss'], atol=1e-5)
    pl.backward(retain_graph=True); vl.backward()                # reference test_action_with_world_model does exactly this
    P = dict(m.named_parameters())
    n = 0
    for k, v in g.items():
        if k.startswith(f'{objective}_grad/'):
            close(P[k.split('/', 1)[1]].grad, v, atol=2e-6, rtol=1e-3); n += 1
        elif k.startswith(f'{objective}_gnorm/'):
            close(P[k.split('/', 1)[1]].grad.norm(), v, atol=1e-6, rtol=1e-3); n += 1
        elif k.startswith(f'{objective}_gsample/'):
            close(P[k.split('/', 1)[1]].grad.flatten()[::97], v, atol=2e-6, rtol=1e-3)
    assert n >= 20
    assert all(p.grad is None for k, p in P.items() if not k.startswith(HEADS)), 'only the two heads learn (D4:5898)'


def test_learning_the_whole_world_model_vs_reference_fixture():
    """learn_full.npz: learn_from_experience(only_learn_policy_value_heads=False) (D4:6045-6075).  The agent embeddings are recomputed by a
    forward WITH gradient through the HIP trunk blocks, the HIP learner returns d loss / d agent_embed next to the head gradients, and
    autograd carries it on: both losses and every one of the reference's 112 / 111 parameter gradients (ppo; norms under pmpo)."""
    g = load_golden('learn_full.npz')
    exp = Experience(latents=t(g['exp_latents']), agent_embed=t(g['exp_agent_embed']), rewards=t(g['exp_rewards']), values=t(g['exp_values']),
                     log_probs=Actions(t(g['exp_log_probs']), None), actions=Actions(t(g['exp_actions']), None), lens=t(g['exp_lens']),
                     terminals=t(g['exp_terminals']), is_truncated=~t(g['exp_terminals']), old_action_unembeds=Actions(t(g['exp_unembeds']), None),
                     step_size=4)
    for obj in ('ppo', 'pmpo'):
        m = golden_model('weights_learn_full.npz').cuda()
        pl, vl = m.learn_from_experience(exp, objective=obj, only_learn_policy_value_heads=False)
        close(pl, g[f'{obj}_policy_loss'], atol=1e-5); close(vl, g[f'{obj}_value_loss'], atol=1e-5)
        P = dict(m.named_parameters())
        pl.backward(retain_graph=True)
        n = 0
        for k, v in g.items():
            if k.startswith(f'{obj}_pgrad/'):
                gr = P[k.split('/', 1)[1]].grad
                assert gr is not None, k
                if obj == 'ppo':
                    close(gr, v, atol=3e-6, rtol=2e-3)
                else:
                    close(gr.norm(), v, atol=3e-6, rtol=2e-3)
                n += 1
        assert n == 112
        if obj == 'ppo':
            m.zero_grad()
            vl.backward()
            n = 0
            for k, v in g.items():
                if k.startswith('ppo_vgrad/'):
                    gr = P[k.split('/', 1)[1]].grad
                    assert gr is not None, k
                    close(gr, v, atol=3e-6, rtol=2e-3); n += 1
            assert n == 111
    # with optimisers: one step of each moves trunk and heads
    m = golden_model('weights_learn_full.npz').cuda()
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    po, vo = torch.optim.SGD(m.parameters(), lr=1e-2), torch.optim.SGD(m.parameters(), lr=1e-2)
    m.learn_from_experience(exp, po, vo, only_learn_policy_value_heads=False)
    moved = [k for k, p in m.named_parameters() if not torch.equal(p.detach(), before[k])]
    assert any(k.startswith('transformer.layers.0') for k in moved) and any(k.startswith('value_head') for k in moved)


@pytest.mark.parametrize('tag,kw', (('nogate', dict(objective='ppo', use_delight_gating=False)), ('temp', dict(objective='spo', delight_temperature=2.5)),
                                    ('rawadv', dict(objective='ppo', normalize_advantages=False)),
                                    ('pmpo_norm', dict(objective='pmpo', normalize_advantages=True, eps=1e-3))))
def test_non_default_learner_options_vs_reference_fixture(tag, kw):
    """use_delight_gating / delight_temperature / normalize_advantages / eps overrides of learn_from_experience."""
    m = golden_model().cuda()
    G, g = load_golden('generate.npz'), load_golden('options.npz')
    pl, vl = m.learn_from_experience(fixture_experience(G), **kw)
    close(pl, g[f'learn_{tag}_policy_loss'], atol=1e-5, rtol=1e-4); close(vl, g[f'learn_{tag}_value_loss'], atol=1e-5)
    pl.backward(retain_graph=True); vl.backward()
    P = dict(m.named_parameters())
    n = 0
    for k, v in g.items():
        if k.startswith(f'learn_{tag}_grad/'):
            ref = t(v)
            close(P[k.split('/', 1)[1]].grad, ref, atol=2e-6 + 1e-4 * ref.abs().max().item(), rtol=1e-3); n += 1
    assert n >= 10


def test_non_default_hyper_parameters_vs_reference_fixture():
    """Head MLP depths, reward / value ranges and bin counts, max_steps, softclamp value, GAE / PPO / PMPO / entropy constants,
    delight gating off, forward KL — through the engine: rollout with early termination, then all three objectives."""
    g = load_golden('hyper.npz')
    m = golden_model('weights_hyper.npz').cuda()
    nz = golden_noise(g, 'cached_')
    e = m.generate(6, batch_size=4, return_for_policy_optimization=True, num_steps=2, noise=nz)
    close(e.latents, g['cached_latents'], atol=1e-4); close(e.values, g['cached_values'], atol=1e-4); close(e.rewards, g['cached_rewards'], atol=1e-4)
    assert np.array_equal(e.actions.discrete.cpu().numpy(), g['cached_actions']) and np.array_equal(e.lens.cpu().numpy(), g['cached_lens'])
    exp = Experience(latents=t(g['cached_latents']), agent_embed=t(g['cached_agent_embed']), rewards=t(g['cached_rewards']),
                     values=t(g['cached_values']), log_probs=Actions(t(g['cached_log_probs']), None), actions=Actions(t(g['cached_actions']), None),
                     lens=t(g['cached_lens']), terminals=t(g['cached_terminals']), is_truncated=~t(g['cached_terminals']),
                     old_action_unembeds=Actions(t(g['cached_unembeds']), None), step_size=8)
    P = dict(m.named_parameters())
    for obj in ('ppo', 'spo', 'pmpo'):
        m.zero_grad()
        pl, vl = m.learn_from_experience(exp, objective=obj)
        close(pl, g[f'{obj}_policy_loss'], atol=1e-5, rtol=1e-4); close(vl, g[f'{obj}_value_loss'], atol=1e-5)
        pl.backward(retain_graph=True); vl.backward()
        for k, v in g.items():
            if k.startswith(f'{obj}_grad/'):
                ref = t(v)
                close(P[k.split('/', 1)[1]].grad, ref, atol=2e-6 + 1e-4 * ref.abs().max().item(), rtol=1e-3)


def test_learn_vs_oracle_with_terminations_and_two_action_types():
    m = small_model(num_discrete_actions=(3, 2))
    from util import randomize_weights
    randomize_weights(m, terminal_bias=-0.5)
    cfg, W = oracle_config(m), oracle_weights(m)
    B, T = 6, 6
    nz = make_noise(cfg, T, B, 31)
    ref = restate.generate(cfg, W, T, batch_size=B, noise=nz)
    assert ref['terminals'].any() and (ref['lens'] < ref['latents'].shape[1]).any()
    m = m.cuda()
    e = m.generate(T, batch_size=B, return_for_policy_optimization=True, noise=nz)
    for obj in ('ppo', 'spo', 'pmpo'):
        Wg = {k: (v.clone().requires_grad_() if k.startswith(HEADS) else v) for k, v in W.items()}
        pl_o, vl_o = restate.learn_losses(cfg, Wg, ref, obj)
        pl_o.backward(); vl_o.backward()
        m.zero_grad()
        pl, vl = m.learn_from_experience(e, objective=obj)
        close(pl, pl_o, atol=2e-5); close(vl, vl_o, atol=2e-5)
        pl.backward(); vl.backward()
        for k, p in m.named_parameters():
            if k.startswith(HEADS) and p.numel() > 0:
                close(p.grad, Wg[k].grad, atol=5e-6, rtol=2e-3)


@pytest.mark.parametrize('i', range(6))
def test_learn_vs_oracle_random_config_sweep(i):
    """Seeded random architectures / batch shapes / objectives (with early terminations so that the masks matter):
    losses and the gradients of both heads, HIP vs the oracle's autograd."""
    import random
    from util import randomize_weights
    rng = random.Random(900 + i)
    kw = dict(dim=rng.choice([32, 64, 96]), attn_heads=rng.choice([1, 2]), depth=rng.choice([1, 2, 3]), time_block_every=rng.choice([1, 2]),
              num_latent_tokens=rng.choice([3, 6]), dim_latent=rng.choice([4, 8]), num_spatial_tokens=rng.choice([1, 2, 4]),
              num_register_tokens=rng.choice([0, 2, 8]), num_discrete_actions=rng.choice([3, (2, 3), (4, 2, 2)]), num_tasks=0,
              multi_token_pred_len=rng.choice([1, 8]))
    m = randomize_weights(small_model(**kw), terminal_bias=rng.choice([-0.5, -2.0]))
    with torch.no_grad():       # logits of O(5), not O(300): fp32 rounding of a logit is then ~1e-6 in the log-prob, and the
        dict(m.named_parameters())['action_embedder.discrete_action_unembed'].mul_(0.02)    # delight gate exp(-lp*adv) stays tame
    cfg, W = oracle_config(m), oracle_weights(m)
    B, T = rng.choice([2, 5, 9]), rng.choice([2, 4, 7])
    nz = make_noise(cfg, T, B, 40 + i)
    ref = restate.generate(cfg, W, T, batch_size=B, noise=nz)
    m = m.cuda()
    # the learner is compared on IDENTICAL inputs (the oracle's own rollout): z-scored advantages over a handful of samples
    # amplify the ~1e-6 rollout differences far beyond any kernel error otherwise
    e = Experience(latents=ref['latents'], agent_embed=ref['agent_embed'], rewards=ref['rewards'], values=ref['values'],
                   log_probs=Actions(ref['log_probs'], None), actions=Actions(ref['actions'], None), lens=ref['lens'],
                   terminals=ref['terminals'], is_truncated=ref['is_truncated'], old_action_unembeds=Actions(ref['old_action_unembeds'], None),
                   step_size=ref['step_size'])
    obj = ('ppo', 'spo', 'pmpo')[i % 3]
    Wg = {k: (v.clone().requires_grad_() if k.startswith(HEADS) else v) for k, v in W.items()}
    pl_o, vl_o = restate.learn_losses(cfg, Wg, ref, obj)
    pl_o.backward(); vl_o.backward()
    pl, vl = m.learn_from_experience(e, objective=obj)
    pl.backward(); vl.backward()
    # Tolerance from the problem's own conditioning.  In the oracle the recomputed log-prob equals the stored one bit for bit
    # (ratio == 1 exactly) and the advantage is a difference of nearly equal returns / values; on the GPU the policy MLP and
    # the GAE accumulate in another order, so those quantities move by ~1e-6 relative — for the log-probs that is 1e-6 of the
    # LOGIT magnitude (hundreds with these random heads, i.e. ratio = 1 +- 2e-4) — which z-scoring over a handful of valid
    # steps (or pmpo's raw advantages) amplifies further.  So the oracle is run a second time on inputs perturbed at that
    # level, and the GPU must stay within 20x of what that perturbation does (+ 1e-5 of the tensor's scale).
    gen = torch.Generator().manual_seed(i)
    jig = lambda x, r: x * (1 + r * (2 * torch.rand(x.shape, generator=gen) - 1))
    ref2 = dict(ref, values=jig(ref['values'], 2e-6), rewards=jig(ref['rewards'], 2e-6), agent_embed=jig(ref['agent_embed'], 2e-6),
                log_probs=ref['log_probs'] + 1e-6 * ref['old_action_unembeds'].abs().max() * (2 * torch.rand(ref['log_probs'].shape, generator=gen) - 1))
    Wp = {k: (v.clone().requires_grad_() if k.startswith(HEADS) else v) for k, v in W.items()}
    pl_p, vl_p = restate.learn_losses(cfg, Wp, ref2, obj)
    pl_p.backward(); vl_p.backward()
    close(pl, pl_o, atol=2e-5 + 20 * abs(pl_p.item() - pl_o.item())); close(vl, vl_o, atol=2e-5 + 20 * abs(vl_p.item() - vl_o.item()))
    for k, p in m.named_parameters():
        if k.startswith(HEADS) and p.numel() > 0:
            sens = (Wp[k].grad - Wg[k].grad).abs().max().item()
            close(p.grad, Wg[k].grad, atol=20 * sens + 1e-5 * Wg[k].grad.abs().max().item() + 1e-9, rtol=1e-4)


def test_three_trainer_steps_vs_reference_fixture():
    """generate -> learn -> clip_grad_norm_(0.5) -> AdamW(3e-4) on each head, natively (trainers.py:1430-1452)."""
    g = load_golden('trainer.npz')
    m = golden_model().cuda()
    _, W0 = golden_oracle()
    tr = DreamTrainer(m, batch_size=4, generate_timesteps=4)
    for step in range(3):
        nz = golden_noise(g, f'step{step}_')
        dreams = m.generate(5, batch_size=4, return_rewards_per_frame=True, return_agent_actions=True,
                            return_log_probs_and_values=True, noise=nz)
        assert np.array_equal(dreams.actions.discrete.cpu().numpy(), g[f'step{step}_actions'])
        losses = tr.learn(dreams)
        close(losses[0], g[f'step{step}_policy_loss'], atol=2e-5); close(losses[1], g[f'step{step}_value_loss'], atol=2e-5)
        close(tr._state['policy']['scratch'][0], g[f'step{step}_policy_gnorm'], atol=1e-5, rtol=1e-3)
        close(tr._state['value']['scratch'][0], g[f'step{step}_value_gnorm'], atol=1e-5, rtol=1e-3)
    P = dict(m.named_parameters())
    for k, v in g.items():
        if k.startswith('final/'):
            close(P[k[6:]], v, atol=5e-6)
        elif k.startswith('final_sample/'):
            close(P[k[13:]].flatten()[::97], v, atol=5e-6)
        elif k.startswith('final_delta_norm/'):
            close((P[k[17:]].detach().cpu() - W0[k[17:]]).norm(), v, atol=1e-5, rtol=1e-2)


def test_dream_trainer_constructor_follows_the_reference():
    """Positional / keyword order of trainers.py:1331-1349; a non-default optimiser class steps both heads through PyTorch
    on the engine's gradients; CPU / tracker options raise instead of being ignored."""
    m = small_model().cuda()
    with pytest.raises(NotImplementedError):
        DreamTrainer(m, batch_size=2, num_train_steps=1, cpu=True)              # the reference test's own call (tests/test_dreamer.py:755)
    with pytest.raises(NotImplementedError):
        DreamTrainer(m, use_wandb=True)
    before = {k: p.detach().clone() for k, p in m.named_parameters() if k.startswith(HEADS)}
    tr = DreamTrainer(m, torch.optim.SGD, 3, 4, 1e-2, objective='spo', num_train_steps=2)     # positional: optim_klass, batch_size, generate_timesteps, lr
    assert tr.batch_size == 3 and tr.generate_timesteps == 4 and tr.lr == 1e-2
    tr()
    assert tr.step == 2
    moved = [k for k, p in m.named_parameters() if k.startswith(HEADS) and p.numel() > 0 and not torch.equal(p, before[k])]
    assert any(k.startswith('policy_head') for k in moved) and any(k.startswith('value_head') for k in moved)
    frozen = [k for k, p in m.named_parameters() if not k.startswith(HEADS) and p.numel() > 0]
    assert frozen                                                   # the trunk is not an optimiser target
    tr2 = DreamTrainer(m, torch.optim.AdamW, batch_size=2, generate_timesteps=3, num_train_steps=1)   # AdamW -> the fused native step
    assert tr2.torch_optims is None
    tr2()


def test_optimizer_arguments_step_the_heads_like_the_reference():
    m = small_model().cuda()
    cfg = oracle_config(m)
    nz = make_noise(cfg, 4, 3, 5)
    e = m.generate(4, batch_size=3, return_for_policy_optimization=True, noise=nz)
    before = [p.detach().clone() for p in m.policy_head_parameters() if p.numel() > 0]
    popt = torch.optim.AdamW([p for p in m.policy_head_parameters() if p.numel() > 0], lr=1e-3)
    vopt = torch.optim.AdamW(m.value_head_parameters(), lr=1e-3)
    pl, vl = m.learn_from_experience(e, policy_optim=popt, value_optim=vopt)
    assert pl.numel() == 1 and vl.numel() == 1
    after = [p for p in m.policy_head_parameters() if p.numel() > 0]
    assert any(not torch.equal(a, b) for a, b in zip(after, before))
    e2 = m.generate(4, batch_size=3, return_for_policy_optimization=True, noise=nz)       # heads are read in place: no re-prepare needed
    assert not torch.equal(e2.values, e.values)


CFG2_ARCH = dict(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, attn_heads=8, attn_dim_head=64, num_spatial_tokens=4,
                 num_register_tokens=8, max_steps=64, multi_token_pred_len=8, num_discrete_actions=4)


@pytest.mark.parametrize('terminal_bias', [-10., -3.])
def test_learn_at_baseline_size_vs_oracle(terminal_bias):
    """The actor/critic step of the headline (BASELINE config 2: dim 512, depth 6, B = 256 trajectories x T = 16 frames = 4096 learner rows —
    bench.py's `actor_critic_step_ms`) against the oracle AT THAT SIZE: learn_from_experience(ppo) = GAE -> z-score -> PPO surrogate + entropy ->
    HL-Gauss CE (D4:5893-6305), both backward passes, then clip_grad_norm_(0.5) + AdamW(3e-4) on each head (trainers.py:1430-1452).  At 4096
    rows the learner's MLP GEMMs take other tile configurations than every dim <= 128 test (k-sliced weight gradients, input gradients
    through a transposed weight image).  Checked, on ONE Experience fed to both sides (the GPU's own B = 256 rollout; terminal bias -10 is
    bench.py's model — nothing terminates; -3 ends a share of the trajectories early so that lens / masks / the terminal bootstrap matter):
      * both losses, every gradient of policy_head.* / value_head.* / discrete_action_unembed, both gradient norms vs restate.learn_losses + autograd;
      * the head weights after ONE DreamTrainer.learn vs torch clip_grad_norm_ + torch.optim.AdamW — on the GPU's gradients elementwise (the
        optimiser kernel at this size), on the oracle's gradients as a relative l2 distance of the update (AdamW's first step is
        lr * g / (|g| + 1e-8): elementwise it amplifies a 1e-9 gradient difference without bound where |g| ~ 1e-8)."""
    from dreamer4_amd import DynamicsWorldModel
    from util import randomize_weights
    torch.set_num_threads(min(16, torch.get_num_threads()))
    torch.manual_seed(0)
    m = randomize_weights(DynamicsWorldModel(**CFG2_ARCH), seed=0, terminal_bias=terminal_bias)
    cfg, W = oracle_config(m), oracle_weights(m)
    B, T = 256, 16
    m = m.cuda()
    nz = make_noise(cfg, T, B, 1234)
    e = m.generate(T, batch_size=B, return_for_policy_optimization=True, num_steps=4, noise=nz)
    assert e.agent_embed.shape == (B, e.latents.shape[1], 512)
    if terminal_bias <= -10.:
        assert e.latents.shape[1] == T and int(e.lens.min()) == T
    else:
        assert 0 < int((e.lens < e.latents.shape[1]).sum()) < B, 'this case is meant to end some, not all, trajectories early'
    cpu = lambda x: x.detach().cpu()
    ref = dict(latents=cpu(e.latents), agent_embed=cpu(e.agent_embed), rewards=cpu(e.rewards), values=cpu(e.values), log_probs=cpu(e.log_probs.discrete),
               actions=cpu(e.actions.discrete), lens=cpu(e.lens), terminals=cpu(e.terminals), is_truncated=cpu(e.is_truncated),
               old_action_unembeds=cpu(e.old_action_unembeds.discrete), step_size=e.step_size)
    Wg = {k: (v.clone().requires_grad_() if k.startswith(HEADS) else v) for k, v in W.items()}
    pl_o, vl_o = restate.learn_losses(cfg, Wg, ref, 'ppo')
    pl_o.backward(); vl_o.backward()
    # the oracle once more on inputs moved by fp32 rounding: what that does to each quantity is the yardstick (as in the random-config sweep above)
    gen = torch.Generator().manual_seed(7)
    jig = lambda x, r: x * (1 + r * (2 * torch.rand(x.shape, generator=gen) - 1))
    ref2 = dict(ref, values=jig(ref['values'], 2e-6), rewards=jig(ref['rewards'], 2e-6), agent_embed=jig(ref['agent_embed'], 2e-6),
                log_probs=ref['log_probs'] + 1e-6 * ref['old_action_unembeds'].abs().max() * (2 * torch.rand(ref['log_probs'].shape, generator=gen) - 1))
    Wp = {k: (v.clone().requires_grad_() if k.startswith(HEADS) else v) for k, v in W.items()}
    pl_p, vl_p = restate.learn_losses(cfg, Wp, ref2, 'ppo')
    pl_p.backward(); vl_p.backward()

    before = {k: p.detach().cpu().clone() for k, p in m.named_parameters() if k.startswith(HEADS) and p.numel() > 0}
    tr = DreamTrainer(m, batch_size=B, generate_timesteps=T - 1, objective='ppo')
    losses = tr.learn(e)                                        # learn_from_experience + both backward passes + clip + AdamW on both heads
    close(losses[0], pl_o, atol=2e-5 + 20 * abs(pl_p.item() - pl_o.item())); close(losses[1], vl_o, atol=2e-5 + 20 * abs(vl_p.item() - vl_o.item()))
    names = {id(p): k for k, p in m.named_parameters()}
    n_checked = 0
    for head, prefix in (('policy', ('policy_head', 'action_embedder.discrete_action_unembed')), ('value', ('value_head',))):
        grp = m._groups[head]
        keys = [names[id(p)] for p in grp['params']]
        assert keys and all(k.startswith(prefix) for k in keys)
        g_gpu = grp['grad'].detach().cpu()
        off, worst = 0, 0.
        for k, p in zip(keys, grp['params']):
            gg, go = g_gpu[off:off + p.numel()].view(p.shape), Wg[k].grad
            sens = (Wp[k].grad - go).abs().max().item()
            close(gg, go, atol=20 * sens + 1e-5 * go.abs().max().item() + 1e-9, rtol=1e-4)
            off += p.numel(); n_checked += 1
        g_or = torch.cat([Wg[k].grad.reshape(-1) for k in keys])
        g_pp = torch.cat([Wp[k].grad.reshape(-1) for k in keys])
        # norms in float64: torch's fp32 `.norm()` of 9.5 M elements on the CPU is itself off by 3e-4 ... 2e-3 relative (its accumulation order
        # depends on the thread count) - more than the whole GPU-vs-oracle difference (measured 2.5e-6 relative on every tensor)
        n_or, n_pp = g_or.double().norm().item(), g_pp.double().norm().item()
        close(tr._state[head]['scratch'][0], n_or, atol=1e-6 + 20 * abs(n_pp - n_or), rtol=2e-5)
        assert (g_gpu.double() - g_or.double()).norm().item() <= 20 * (g_pp.double() - g_or.double()).norm().item() + 1e-5 * n_or

        def torch_step(grads):                                  # trainers.py:1436-1452 on plain torch
            ps = [before[k].clone().requires_grad_() for k in keys]
            opt = torch.optim.AdamW(ps, lr=3e-4, weight_decay=0.)
            o = 0
            for q in ps:
                q.grad = grads[o:o + q.numel()].view(q.shape).clone(); o += q.numel()
            torch.nn.utils.clip_grad_norm_(ps, 0.5)
            opt.step()
            return torch.cat([q.detach().reshape(-1) for q in ps])
        w0 = torch.cat([before[k].reshape(-1) for k in keys])
        w_gpu = grp['flat'].detach().cpu()
        assert not torch.equal(w_gpu, w0)
        w_t = torch_step(g_gpu)                                 # the optimiser kernel on ITS gradients: elementwise
        tiny = g_gpu.abs() < 1e-6 * g_gpu.abs().max()           # (|g| within ~100 x eps of 0: the step there is ill-conditioned in g's last bits)
        assert (w_gpu - w_t)[~tiny].abs().max().item() <= 2e-7 + 1e-6 * w0.abs().max().item()
        assert (w_gpu - w_t).abs().max().item() <= 3.1e-4        # never more than one full step of lr
        w_o = torch_step(g_or)                                  # ... and on the oracle's gradients: the update as a whole
        d = lambda x: x.double()
        rel = (d(w_gpu) - d(w_o)).norm().item() / (d(w_o) - d(w0)).norm().item()
        rel_p = (d(torch_step(g_pp)) - d(w_o)).norm().item() / (d(w_o) - d(w0)).norm().item()
        assert rel <= 20 * rel_p + 1e-3, (head, rel, rel_p)
    assert n_checked >= 10
