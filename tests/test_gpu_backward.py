"""GPU: the first slice of the trunk backward (SURVEY.md 8f-3 groundwork) — FeedForward and within-frame Attention blocks as
forward + backward HIP operators (include/d4hip.h d4_ff_* / d4_space_attn_*), checked against autograd of the oracle's restatement
(oracle/restate.py feedforward / attention, evaluated in float64).  Tolerance: 2e-4 of each tensor's scale (fp32 MFMA accumulation
over up to a few thousand rows)."""
import pytest
import torch

from dreamer4_amd import trunk_ops
from oracle import restate

pytestmark = pytest.mark.gpu


def close(a, b, name, tol=2e-4):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f'{name}: max abs diff {err:.3e} vs scale {scale:.3e}'


def _ff_params(D, inner, g):
    r = lambda *s, k=1.: torch.randn(*s, generator=g) * k
    return {'norm.weight': 1. + r(D, k=.1), 'proj_in.weight': r(2 * inner, D, k=D ** -.5), 'proj_in.bias': r(2 * inner, k=.3),
            'proj_out.weight': r(D, inner, k=inner ** -.5), 'proj_out.bias': r(D, k=.3)}


@pytest.mark.parametrize('lead,D,inner', [((37,), 64, 170), ((3, 100), 512, 1365), ((1,), 32, 85)])
def test_feedforward_forward_and_backward_vs_oracle_autograd(lead, D, inner):
    g = torch.Generator().manual_seed(3)
    W = _ff_params(D, inner, g)
    x = torch.randn(*lead, D, generator=g) * 1.5
    dy = torch.randn(*lead, D, generator=g)
    Wd = {k: v.double().requires_grad_() for k, v in W.items()}
    xd = x.double().requires_grad_()
    ref = restate.feedforward(Wd, '', xd)
    ref.backward(dy.double())
    Wg = {k: v.cuda().requires_grad_() for k, v in W.items()}
    xg = x.cuda().requires_grad_()
    y = trunk_ops.feedforward(xg, Wg['norm.weight'], Wg['proj_in.weight'], Wg['proj_in.bias'], Wg['proj_out.weight'], Wg['proj_out.bias'])
    close(y, ref, 'y')
    y.backward(dy.cuda())
    close(xg.grad, xd.grad, 'dx')
    for k in W:
        close(Wg[k].grad, Wd[k].grad, 'd ' + k)


def _attn_params(D, heads, dh, g):
    r = lambda *s, k=1.: torch.randn(*s, generator=g) * k
    hd = heads * dh
    return {'norm.weight': 1. + r(D, k=.1), 'to_q.weight': r(hd, D, k=3. * D ** -.5), 'to_k.weight': r(hd, D, k=D ** -.5), 'to_v.weight': r(hd, D, k=D ** -.5),
            'to_out.weight': r(D, hd, k=hd ** -.5), 'to_gates.0.weight': r(heads, D, k=D ** -.5), 'k_heads_rmsnorm.gamma': r(heads, dh, k=.3),
            'to_learned_value_residual_mix.0.weight': r(heads, D, k=D ** -.5), 'to_learned_value_residual_mix.0.bias': r(heads, k=.5)}


@pytest.mark.parametrize('F_,S,D,heads,dh,has_rv,ns,clamp,belief', [
    (5, 9, 64, 2, 64, True, 1, 50., True),            # the dynamics trunk's form (value residual, one special token)
    (3, 30, 128, 3, 32, False, 6, 50., True),         # first layer (no residual), tokenizer-encoder-like special block, heads not a multiple of 4
    (4, 12, 64, 5, 16, True, 0, 2., False),           # tight soft clamp, no belief projection, no special tokens
    (130, 15, 512, 8, 64, True, 1, 50., True),        # BASELINE cfg 2 geometry (15 tokens per frame, 8 x 64 heads)
    (3, 64, 64, 2, 64, True, 2, 50., True),           # the operator's limit: 64 tokens per frame (LDS arrays at capacity 64)
    (2, 41, 64, 2, 32, True, 1, 50., True),
])
def test_space_attention_forward_and_backward_vs_oracle_autograd(F_, S, D, heads, dh, has_rv, ns, clamp, belief):
    g = torch.Generator().manual_seed(7)
    W = _attn_params(D, heads, dh, g)
    x = torch.randn(F_, S, D, generator=g) * 1.5
    rv = torch.randn(F_, S, heads, dh, generator=g) if has_rv else None
    dy = torch.randn(F_, S, D, generator=g)
    Wd = {k: v.double().requires_grad_() for k, v in W.items()}
    xd = x.double().requires_grad_()
    rvd = rv.double().requires_grad_() if has_rv else None
    mask = restate.special_token_mask(S, ns) if ns > 0 else None
    ref, _ = restate.attention(Wd, '', xd, heads=heads, dim_head=dh, residual_values=rvd, softclamp_value=clamp, mask=mask, belief=belief)
    ref.backward(dy.double())
    Wg = {k: v.cuda().requires_grad_() for k, v in W.items()}
    xg = x.cuda().requires_grad_()
    rvg = rv.cuda().requires_grad_() if has_rv else None
    y = trunk_ops.space_attention(xg, Wg['norm.weight'], Wg['to_q.weight'], Wg['to_k.weight'], Wg['to_v.weight'], Wg['to_out.weight'],
                                  Wg['to_gates.0.weight'], Wg['k_heads_rmsnorm.gamma'], residual_values=rvg,
                                  mix_weight=Wg['to_learned_value_residual_mix.0.weight'] if has_rv else None,
                                  mix_bias=Wg['to_learned_value_residual_mix.0.bias'] if has_rv else None,
                                  softclamp_value=clamp, num_special=ns, belief=belief)
    close(y, ref, 'y')
    y.backward(dy.cuda())
    close(xg.grad, xd.grad, 'dx')
    if has_rv:
        close(rvg.grad, rvd.grad, 'd residual_values')
    for k in W:
        if 'value_residual_mix' in k and not has_rv:
            continue
        close(Wg[k].grad, Wd[k].grad, 'd ' + k)


@pytest.mark.parametrize('B,T,S,D,heads,dh,has_rv,clamp', [(2, 7, 5, 64, 2, 64, True, 50.), (1, 32, 3, 64, 3, 32, False, 50.), (3, 16, 15, 128, 2, 16, True, 3.), (1, 64, 2, 64, 2, 64, True, 50.), (2, 48, 3, 64, 1, 32, False, 50.)])
def test_time_attention_forward_and_backward_vs_oracle_autograd(B, T, S, D, heads, dh, has_rv, clamp):
    from einops import rearrange
    g = torch.Generator().manual_seed(9)
    W = _attn_params(D, heads, dh, g)
    x = torch.randn(B, T, S, D, generator=g) * 1.5
    rv = torch.randn(B, T, S, heads, dh, generator=g) if has_rv else None
    dy = torch.randn(B, T, S, D, generator=g)
    inv_freq = 1.0 / (10000. ** (torch.arange(0, dh, 2).float() / dh))
    Wd = {k: v.double().requires_grad_() for k, v in W.items()}
    xd = x.double().requires_grad_()
    rvd = rv.double().requires_grad_() if has_rv else None
    # the reference runs the time layers on 'b t s d -> (b s) t d' (dreamer4.py:3178), causal, rotary positions 0..T-1
    rot = restate.rotary_freqs(restate.Config(dim=D, dim_latent=4, num_latent_tokens=1, attn_dim_head=dh), T, 0, inv_freq.double())
    ref, _ = restate.attention(Wd, '', rearrange(xd, 'b t s d -> (b s) t d'), heads=heads, dim_head=dh, rot=rot, causal=True,
                               residual_values=rearrange(rvd, 'b t s h d -> (b s) t h d') if has_rv else None, softclamp_value=clamp)
    ref = rearrange(ref, '(b s) t d -> b t s d', b=B)
    ref.backward(dy.double())
    Wg = {k: v.cuda().requires_grad_() for k, v in W.items()}
    xg = x.cuda().requires_grad_()
    rvg = rv.cuda().requires_grad_() if has_rv else None
    y = trunk_ops.time_attention(xg, Wg['norm.weight'], Wg['to_q.weight'], Wg['to_k.weight'], Wg['to_v.weight'], Wg['to_out.weight'],
                                 Wg['to_gates.0.weight'], Wg['k_heads_rmsnorm.gamma'], inv_freq.cuda(), residual_values=rvg,
                                 mix_weight=Wg['to_learned_value_residual_mix.0.weight'] if has_rv else None,
                                 mix_bias=Wg['to_learned_value_residual_mix.0.bias'] if has_rv else None, softclamp_value=clamp)
    close(y, ref, 'y')
    y.backward(dy.cuda())
    close(xg.grad, xd.grad, 'dx')
    if has_rv:
        close(rvg.grad, rvd.grad, 'd residual_values')
    for k in W:
        if 'value_residual_mix' in k and not has_rv:
            continue
        close(Wg[k].grad, Wd[k].grad, 'd ' + k)


@pytest.mark.parametrize('G,nq,nk,D,Dc,heads,dh,item_major,ctx_norm,clamp', [
    (37, 1, 7, 64, 64, 4, 64, True, True, None),          # AttentionPool: one query per token row over the stack of 7 hiddens
    (6, 3, 20, 64, 64, 2, 32, False, True, None),         # special tokens over the ordinary tokens of their frame
    (5, 4, 64, 128, 8, 3, 16, False, True, 5.),           # learned-query pool shape: 64 keys of a narrow context, soft clamp
    (9, 64, 5, 64, 32, 2, 64, False, False, None),        # many queries, context not normalised
])
def test_cross_attention_forward_and_backward_vs_oracle_autograd(G, nq, nk, D, Dc, heads, dh, item_major, ctx_norm, clamp):
    g = torch.Generator().manual_seed(13)
    r = lambda *s_, k=1.: torch.randn(*s_, generator=g) * k
    hd = heads * dh
    W = {'norm.weight': 1. + r(D, k=.1), 'norm_context.weight': 1. + r(Dc, k=.1), 'to_q.weight': r(hd, D, k=3. * D ** -.5), 'to_k.weight': r(hd, Dc, k=Dc ** -.5),
         'to_v.weight': r(hd, Dc, k=Dc ** -.5), 'to_out.weight': r(D, hd, k=hd ** -.5), 'to_gates.0.weight': r(heads, D, k=D ** -.5),
         'k_heads_rmsnorm.gamma': r(heads, dh, k=.3)}
    q = r(G, nq, D, k=1.5)
    c = r(G, nk, Dc, k=1.5)
    dy = r(G, nq, D)
    Wd = {k: v.double().requires_grad_() for k, v in W.items()}
    qd, cd = q.double().requires_grad_(), c.double().requires_grad_()
    ref, _ = restate.attention(Wd, '', qd, heads=heads, dim_head=dh, context=cd, belief=True, has_ctx_norm=ctx_norm, softclamp_value=clamp)
    ref.backward(dy.double())
    Wg = {k: v.cuda().requires_grad_() for k, v in W.items()}
    qg = q.cuda().requires_grad_()
    cg = c.cuda().requires_grad_()
    cin = cg.transpose(0, 1).contiguous() if item_major else cg            # (nk, G, Dc) for the stack-of-hiddens layout
    y = trunk_ops.cross_attention(qg, cin, Wg['norm.weight'], Wg['norm_context.weight'] if ctx_norm else None, Wg['to_q.weight'], Wg['to_k.weight'],
                                  Wg['to_v.weight'], Wg['to_out.weight'], Wg['to_gates.0.weight'], Wg['k_heads_rmsnorm.gamma'],
                                  context_item_major=item_major, softclamp_value=clamp)
    close(y, ref, 'y')
    y.backward(dy.cuda())
    close(qg.grad, qd.grad, 'dq_tokens'); close(cg.grad, cd.grad, 'dcontext')
    for k in W:
        if k == 'norm_context.weight' and not ctx_norm:
            continue
        close(Wg[k].grad, Wd[k].grad, 'd ' + k)


@pytest.mark.parametrize('kw,b,t', [(dict(dim=64, dim_latent=8, num_latent_tokens=4, depth=4, time_block_every=2, attn_heads=2, attn_dim_head=32, num_discrete_actions=4), 2, 5),
                                    (dict(dim=128, dim_latent=16, num_latent_tokens=8, num_spatial_tokens=4, depth=5, time_block_every=4, attn_heads=2, attn_dim_head=64,
                                          num_discrete_actions=4), 3, 4),
                                    # BASELINE config 2's architecture at the shape bench.py's train_flow_step times (15 tokens per frame, 16 frames)
                                    (dict(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, attn_heads=8, attn_dim_head=64, num_spatial_tokens=4,
                                          num_register_tokens=8, num_discrete_actions=4), 2, 16)])
@pytest.mark.parametrize('save_forward', ['1', '0'])
def test_trunk_forward_and_backward_vs_oracle_autograd(kw, b, t, save_forward, monkeypatch):
    """The whole AxialSpaceTimeTransformer (dreamer4.py:2927-3267) as a composition of the HIP forward + backward blocks, against float64
    autograd of the oracle's `transformer`: output, d tokens and the gradient of every trunk parameter — with the blocks keeping their
    forward workspace (`*_backward_saved`, the default) and with the backward recomputing the forward (D4_TRUNK_SAVE_FORWARD=0)."""
    monkeypatch.setenv('D4_TRUNK_SAVE_FORWARD', save_forward)
    if kw['dim'] == 512 and save_forward == '0':
        pytest.skip('the recompute form is covered at the small shapes')
    # the recompute runs also take the dispatcher route (torch.ops.d4hip.swiglu_ff / attn_block_*), the saved ones the autograd.Function route
    monkeypatch.setenv('D4_TRUNK_DISPATCHER', '1' if (save_forward == '0' or kw['dim'] == 128) else '0')
    from dreamer4_amd import DynamicsWorldModel
    from util import oracle_config, randomize_weights
    torch.manual_seed(1)
    m = randomize_weights(DynamicsWorldModel(**kw))
    cfg = oracle_config(m)
    W = {k: v.detach().clone() for k, v in m.state_dict().items() if k.startswith('transformer.')}
    s = 1 + cfg.num_spatial_tokens + cfg.num_register_tokens + 1 + 1            # flow | spatial | registers | action | agent
    g = torch.Generator().manual_seed(2)
    tokens = torch.randn(b, t, s, cfg.dim, generator=g)
    dy = torch.randn(b, t, s, cfg.dim, generator=g)
    isf = lambda k: W[k].is_floating_point() and 'inv_freq' not in k
    Wd = {k: (v.double().requires_grad_() if isf(k) else v.double()) for k, v in W.items()}
    xd = tokens.double().requires_grad_()
    ref, _ = restate.transformer(cfg, Wd, xd)
    ref.backward(dy.double())
    Wg = {k: (v.cuda().requires_grad_() if isf(k) else v.cuda()) for k, v in W.items()}
    xg = tokens.cuda().requires_grad_()
    y = trunk_ops.transformer(Wg, xg, is_time=cfg.is_time, softclamp_value=cfg.attn_softclamp_value)
    close(y, ref, 'trunk output')
    y.backward(dy.cuda())
    close(xg.grad, xd.grad, 'd tokens', tol=5e-4)
    checked = 0
    for k in W:
        if isf(k) and Wd[k].grad is not None:
            assert Wg[k].grad is not None, k
            close(Wg[k].grad, Wd[k].grad, 'd ' + k, tol=5e-4)
            checked += 1
    assert checked >= 20 * cfg.depth


@pytest.mark.parametrize('name', ['shortcut', 'plain'])
def test_dynamics_flow_and_shortcut_losses_vs_reference_fixture(name):
    """train.npz (frozen from the reference's training forward under its own recorded draws): flow + shortcut losses through the HIP
    trunk, and the gradient of their sum with respect to every parameter on the path."""
    from util import golden_oracle, load_golden, t
    g = load_golden('train.npz')
    cfg, W = golden_oracle('weights_train.npz')
    keys = [k[len(name) + 6:] for k in g if k.startswith(name + '_grad/')]
    Wg = {k: (v.cuda().requires_grad_() if k in keys else v.cuda()) for k, v in W.items()}
    cu = lambda a: t(a).cuda()
    fl, sl = trunk_ops.dynamics_flow_losses(
        Wg, cu(g['latents']), cu(g[name + '_noise']), cu(g[name + '_signal_levels']), cu(g[name + '_step_sizes_log2']), name == 'shortcut',
        max_steps=cfg.max_steps, is_time=cfg.is_time, num_spatial_tokens=cfg.num_spatial_tokens, num_register_tokens=cfg.num_register_tokens,
        num_discrete_actions=cfg.num_discrete_actions, discrete_actions=cu(g['actions']), softclamp_value=cfg.attn_softclamp_value)
    close(fl, t(g[name + '_flow_loss']), 'flow loss', tol=1e-5)
    assert abs(sl.item() - float(g[name + '_shortcut_loss'])) <= 1e-5 * max(float(g[name + '_shortcut_loss']), 1e-3)
    (fl + sl).backward()
    for k in keys:
        assert Wg[k].grad is not None, k
        close(Wg[k].grad, t(g[f'{name}_grad/{k}']), 'd ' + k, tol=1e-3)
    assert len(keys) >= 90


def test_world_model_training_forward_matches_the_fixture_and_trains():
    """DynamicsWorldModel.forward without signal levels = the training branch (dreamer4.py:6956-7003, 7335-7431, 7708): total loss against
    the reference fixture with its draws injected, gradients on the mirror's own parameters, then a few AdamW steps on fresh draws
    reduce the loss (the backward is usable, not just correct)."""
    from util import golden_model, load_golden, t
    g = load_golden('train.npz')
    m = golden_model('weights_train.npz').cuda()
    draws = dict(shortcut_train=True, step_sizes_log2=t(g['shortcut_step_sizes_log2']), signal_levels=t(g['shortcut_signal_levels']), noise=t(g['shortcut_noise']))
    total, (fl, sl, *_) = m(latents=t(g['latents']), discrete_actions=t(g['actions']), return_all_losses=True, draws=draws, add_autoregressive_action_loss=False)
    close(fl, t(g['shortcut_flow_loss']), 'flow', tol=1e-5)
    close(total, t(g['shortcut_flow_loss']) + t(g['shortcut_shortcut_loss']), 'total', tol=1e-5)
    total.backward()
    own = dict(m.named_parameters())
    n = 0
    for k in g:
        if k.startswith('shortcut_grad/'):
            close(own[k[14:]].grad, t(g[k]), 'd ' + k[14:], tol=1e-3); n += 1
    assert n >= 90
    with pytest.raises(NotImplementedError):
        m(latents=t(g['latents']), proprio=torch.zeros(3, 4, 2))
    # a short optimisation on a fixed batch of "data" latents, fresh draws every step
    trunk = [p for k, p in m.named_parameters() if p.grad is not None]
    opt = torch.optim.AdamW(trunk, lr=3e-3, weight_decay=0.)
    gen = torch.Generator(device='cuda').manual_seed(3)
    lat = t(g['latents']).cuda()
    # judged on FIXED draws before and after (the loss of a training step depends on its own signal levels, noise and shortcut coin)
    B_, T_ = lat.shape[:2]
    fixed = dict(shortcut_train=False, step_sizes_log2=torch.zeros(B_, dtype=torch.long), signal_levels=torch.randint(0, m.max_steps, (B_, T_), generator=torch.Generator().manual_seed(5)),
                 noise=torch.randn(lat.shape, generator=torch.Generator().manual_seed(6)))
    probe = lambda: m(latents=lat, discrete_actions=t(g['actions']), draws=fixed, add_autoregressive_action_loss=False).item()
    with torch.no_grad():
        first = probe()
    for step in range(40):
        opt.zero_grad(set_to_none=True)
        loss = m(latents=lat, discrete_actions=t(g['actions']), generator=gen, add_autoregressive_action_loss=False)
        loss.backward()
        opt.step()
        m.invalidate_prepared()
    with torch.no_grad():
        last = probe()
    assert last < 0.85 * first, (first, last)


def test_world_model_training_forward_with_rewards_terminals_and_actions_vs_reference_fixture():
    """train_agent.npz: the reference's whole training forward (rewards, terminals, two discrete action types, multi-token prediction 2)
    through the mirror: every loss term, the total, and the gradient of the total on 134 parameters."""
    from util import golden_model, load_golden, t
    g = load_golden('train_agent.npz')
    m = golden_model('weights_train_agent.npz').cuda()
    draws = dict(shortcut_train=True, step_sizes_log2=t(g['step_sizes_log2']), signal_levels=t(g['signal_levels']), noise=t(g['noise']))
    total, L = m(latents=t(g['latents']), discrete_actions=t(g['actions']), rewards=t(g['rewards']), terminals=t(g['terminals']),
                 return_all_losses=True, draws=draws)
    close(L.flow, t(g['flow_loss']), 'flow', tol=1e-5); close(L.rewards, t(g['rewards_loss']), 'rewards', tol=1e-5)
    close(L.terminals, t(g['terminals_loss']), 'terminals', tol=1e-5); close(L.discrete_actions, t(g['discrete_actions_loss']), 'actions', tol=1e-5)
    close(total, t(g['total']), 'total', tol=1e-5)
    total.backward()
    own = dict(m.named_parameters())
    n = 0
    for k in g:
        if k.startswith('grad/'):
            assert own[k[5:]].grad is not None, k
            close(own[k[5:]].grad, t(g[k]), 'd ' + k[5:], tol=1e-3); n += 1
    assert n >= 130
    # rewards / terminals given without their first frame are left-padded as the reference does (dreamer4.py:6905-6911)
    total2 = m(latents=t(g['latents']), discrete_actions=t(g['actions']), rewards=t(g['rewards'])[:, 1:], terminals=t(g['terminals'])[:, 1:], draws=draws)
    close(total2, total, 'total with t-1 rewards', tol=1e-6)
    # variable lengths (dreamer4.py:7418-7426): frames past `lens` drop out of every term
    tl, Ll = m(latents=t(g['latents']), discrete_actions=t(g['actions']), rewards=t(g['rewards']), terminals=t(g['terminals']), lens=t(g['lens']),
               return_all_losses=True, draws=draws)
    terms = torch.cat([Ll.flow.reshape(1), Ll.shortcut.reshape(1), Ll.rewards, Ll.terminals.reshape(1), Ll.discrete_actions])
    close(terms, t(g['lens_terms']), 'terms with lens', tol=1e-5); close(tl, t(g['lens_total']), 'total with lens', tol=1e-5)


def test_world_model_training_forward_with_loss_normalisation_vs_reference_fixture():
    """train_agent.npz norm* keys: `use_loss_normalization=True`, two consecutive calls with the EMA updating."""
    from util import golden_model, load_golden, t
    g = load_golden('train_agent.npz')
    m = golden_model('weights_train_agent.npz', use_loss_normalization=True).cuda()
    draws = dict(shortcut_train=True, step_sizes_log2=t(g['step_sizes_log2']), signal_levels=t(g['signal_levels']), noise=t(g['noise']))
    for call in range(2):
        total, L = m(latents=t(g['latents']), discrete_actions=t(g['actions']), rewards=t(g['rewards']), terminals=t(g['terminals']),
                     return_all_losses=True, draws=draws, update_loss_ema=True)
        terms = torch.cat([L.flow.reshape(1), L.shortcut.reshape(1), L.rewards, L.terminals.reshape(1), L.discrete_actions])
        close(terms, t(g[f'norm{call}_terms']), f'terms of call {call}', tol=2e-5); close(total.reshape(1), t(g[f'norm{call}_total']), 'total', tol=2e-5)
    close(m.reward_loss_normalizer.exp_avg_sq, t(g['norm_state_rewards']), 'running mean squares', tol=1e-4)
    total.backward()                                                 # the normalised total is differentiable like the plain one
    assert m.get_parameter('transformer.layers.0.2.fn.to_q.weight').grad.abs().max().item() > 0


def test_world_model_training_forward_with_continuous_action_cloning_vs_reference_fixture():
    """train_cont.npz: discrete + continuous (Beta) behaviour cloning through the mirror: terms, total, 103 parameter gradients."""
    from util import golden_model, load_golden, t
    g = load_golden('train_cont.npz')
    m = golden_model('weights_train_cont.npz').cuda()
    draws = dict(shortcut_train=False, step_sizes_log2=t(g['step_sizes_log2']), signal_levels=t(g['signal_levels']), noise=t(g['noise']))
    total, L = m(latents=t(g['latents']), discrete_actions=t(g['actions']), continuous_actions=t(g['actions_cont']), return_all_losses=True, draws=draws)
    close(L.flow, t(g['flow_loss']), 'flow', tol=1e-5); close(L.discrete_actions, t(g['discrete_actions_loss']), 'discrete', tol=1e-5)
    close(L.continuous_actions, t(g['continuous_actions_loss']), 'continuous', tol=1e-5); close(total, t(g['total']), 'total', tol=1e-5)
    total.backward()
    own = dict(m.named_parameters())
    n = 0
    for k in g:
        if k.startswith('grad/'):
            close(own[k[5:]].grad, t(g[k]), 'd ' + k[5:], tol=1e-3); n += 1
    assert n >= 100


def test_blocks_compose_with_torch_autograd():
    """x + attention(x), then x + feedforward(x), then a torch loss: gradients flow through both HIP blocks and torch ops."""
    g = torch.Generator().manual_seed(11)
    D, heads, dh, inner = 64, 2, 32, 170
    Wa, Wf = _attn_params(D, heads, dh, g), _ff_params(D, inner, g)
    x = torch.randn(6, 10, D, generator=g)
    rv = torch.randn(6, 10, heads, dh, generator=g)

    def run(x, rv, Wa, Wf, dev):
        if dev == 'cuda':
            a = trunk_ops.space_attention(x, Wa['norm.weight'], Wa['to_q.weight'], Wa['to_k.weight'], Wa['to_v.weight'], Wa['to_out.weight'],
                                          Wa['to_gates.0.weight'], Wa['k_heads_rmsnorm.gamma'], residual_values=rv,
                                          mix_weight=Wa['to_learned_value_residual_mix.0.weight'], mix_bias=Wa['to_learned_value_residual_mix.0.bias'])
            h = x + a
            y = h + trunk_ops.feedforward(h, Wf['norm.weight'], Wf['proj_in.weight'], Wf['proj_in.bias'], Wf['proj_out.weight'], Wf['proj_out.bias'])
        else:
            a, _ = restate.attention(Wa, '', x, heads=heads, dim_head=dh, residual_values=rv, softclamp_value=50., mask=restate.special_token_mask(10, 1))
            h = x + a
            y = h + restate.feedforward(Wf, '', h)
        return (y.tanh() ** 2).mean()

    leaf = lambda t, dev: (t.double() if dev == 'cpu' else t.cuda()).requires_grad_()
    outs = {}
    for dev in ('cpu', 'cuda'):
        xa, rva = leaf(x, dev), leaf(rv, dev)
        Wal, Wfl = {k: leaf(v, dev) for k, v in Wa.items()}, {k: leaf(v, dev) for k, v in Wf.items()}
        loss = run(xa, rva, Wal, Wfl, dev)
        loss.backward()
        outs[dev] = (loss, xa.grad, rva.grad, Wal, Wfl)
    close(outs['cuda'][0], outs['cpu'][0], 'loss', tol=1e-5)
    close(outs['cuda'][1], outs['cpu'][1], 'dx'); close(outs['cuda'][2], outs['cpu'][2], 'd rv')
    for k in Wa:
        close(outs['cuda'][3][k].grad, outs['cpu'][3][k].grad, 'attn d ' + k)
    for k in Wf:
        close(outs['cuda'][4][k].grad, outs['cpu'][4][k].grad, 'ff d ' + k)


def test_argument_errors_are_loud():
    from dreamer4_amd._lib import D4Error
    W = _attn_params(64, 2, 64, torch.Generator().manual_seed(0))
    Wg = {k: v.cuda() for k, v in W.items()}
    with pytest.raises(D4Error, match='items per group'):
        trunk_ops.space_attention(torch.zeros(1, 70, 64, device='cuda'), Wg['norm.weight'], Wg['to_q.weight'], Wg['to_k.weight'], Wg['to_v.weight'],
                                  Wg['to_out.weight'], Wg['to_gates.0.weight'], Wg['k_heads_rmsnorm.gamma'])
    with pytest.raises(D4Error, match='no CPU fallback'):
        trunk_ops.feedforward(torch.zeros(2, 64), *[v for v in _ff_params(64, 170, torch.Generator().manual_seed(0)).values()])
