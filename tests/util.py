"""Shared helpers for the parity tests: build a product model + the matching oracle inputs."""
import torch

from oracle.restate import Config


def oracle_config(m) -> Config:
    return Config(
        dim=m.dim, dim_latent=m.dim_latent, num_latent_tokens=m.num_latent_tokens, depth=m.depth,
        time_block_every=m.time_block_every, attn_heads=m.attn_heads, attn_dim_head=m.attn_dim_head,
        attn_softclamp_value=m.attn_softclamp_value, num_spatial_tokens=m.num_spatial_tokens,
        num_register_tokens=m.num_register_tokens, max_steps=m.max_steps, num_tasks=m.num_tasks,
        num_discrete_actions=m.num_discrete_actions, multi_token_pred_len=m.multi_token_pred_len,
        policy_head_mlp_depth=m.policy_head_mlp_depth, value_head_mlp_depth=m.value_head_mlp_depth,
        terminal_mlp_depth=m.terminal_mlp_depth, predict_terminals=m.predict_terminals,
        reward_range=m.reward_range, reward_num_bins=m.reward_num_bins, value_range=m.value_range,
        value_num_bins=m.value_num_bins, gae_discount_factor=m.gae_discount_factor, gae_lambda=m.gae_lambda,
        ppo_eps_clip=m.ppo_eps_clip, policy_entropy_weight=m.policy_entropy_weight,
        use_delight_gating=m.use_delight_gating, delight_temperature=m.delight_temperature,
        pmpo_pos_to_neg_weight=m.pmpo_pos_to_neg_weight, pmpo_reverse_kl=m.pmpo_reverse_kl,
        pmpo_kl_div_loss_weight=m.pmpo_kl_div_loss_weight,
        num_continuous_actions=getattr(m, 'num_continuous_actions', 0), head_mlp_recipe=getattr(m, 'head_mlp_recipe', 'pre_rms'),
        continuous_beta_param=getattr(m, 'continuous_beta_param', 'softplus_p1'),
        reward_encoder_type=getattr(m, 'reward_encoder_type', 'hl_gauss'),
    )


from dreamer4_amd.synthetic import randomize_weights  # noqa: E402,F401


def oracle_weights(m):
    W = {k: v.detach().cpu().float().clone() for k, v in m.state_dict().items()}
    return W


def make_noise(cfg: Config, frames, batch, seed):
    g = torch.Generator().manual_seed(seed)
    n, dl, A = cfg.num_latent_tokens, cfg.dim_latent, cfg.total_discrete_actions
    nz = dict(
        latent=torch.randn(frames, batch, n, dl, generator=g),
        context=torch.randn(frames, batch, n, dl, generator=g),
        gumbel_u=torch.rand(frames, batch, A, generator=g).clamp(1e-6, 1. - 1e-6),
        bern_u=torch.rand(frames, batch, generator=g),
    )
    nc = getattr(cfg, 'num_continuous_actions', 0)
    if nc > 0:      # Beta sampling as a ratio of Marsaglia-Tsang gammas: (normal, uniform) per rejection round   (drawn last: older fixtures keep their draws)
        nrm = torch.randn(frames, batch, nc, 2, 6, generator=g)
        uni = torch.rand(frames, batch, nc, 2, 6, generator=g).clamp(1e-6, 1. - 1e-6)
        nz['beta'] = torch.stack((nrm, uni), dim=-1)
    return nz


def small_model(**over):
    from dreamer4_amd import DynamicsWorldModel
    kw = dict(dim=64, dim_latent=8, num_latent_tokens=6, depth=4, time_block_every=2, attn_heads=2,
              num_discrete_actions=4, num_tasks=3)
    kw.update(over)
    torch.manual_seed(0)
    return randomize_weights(DynamicsWorldModel(**kw))


# ----------------------------------------------------------------------------- golden fixtures
import os
import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_config_kwargs(weights='weights.npz'):
    w = load_golden(weights)
    kw = {}
    for k, v in w.items():
        if k.startswith('cfg_'):
            v = v.tolist()
            kw[k[4:]] = tuple(v) if isinstance(v, list) else v
    return kw


def golden_oracle(weights='weights.npz'):
    """(Config, W) of the fixture model for oracle/restate.py."""
    w = load_golden(weights)
    cfg = Config(**golden_config_kwargs(weights))
    W = {k: torch.from_numpy(v) for k, v in w.items() if not k.startswith(('cfg_', 'meta_'))}
    return cfg, W


def golden_model(weights='weights.npz', **extra):
    """Product model carrying the fixture weights (loaded by reference state_dict key)."""
    from dreamer4_amd import DynamicsWorldModel
    kw = golden_config_kwargs(weights)
    renc, venc = dict(num_bins=kw.pop('reward_num_bins')), dict(num_bins=kw.pop('value_num_bins'))
    kw.pop('cfg_note', None)
    if 'reward_range' in kw:
        renc['reward_range'] = tuple(kw.pop('reward_range'))
    if 'value_range' in kw:
        venc['reward_range'] = tuple(kw.pop('value_range'))
    kw['num_discrete_actions'] = tuple(kw['num_discrete_actions']) if isinstance(kw['num_discrete_actions'], (tuple, list)) else kw['num_discrete_actions']
    for key in ('head_mlp_recipe', 'reward_encoder_type', 'continuous_beta_param'):
        if key in kw:
            kw[key] = str(kw[key])
    kw = {k: (bool(v) if isinstance(v, (bool, np.bool_)) else v) for k, v in kw.items()}
    m = DynamicsWorldModel(**kw, reward_encoder_kwargs=renc, value_encoder_kwargs=venc, **extra)
    _, W = golden_oracle(weights)
    own = {k: p for k, p in m.named_parameters() if not k.startswith('video_tokenizer.')}       # (a nested tokenizer comes with its own fixture weights)
    missing = [k for k, p in own.items() if p.numel() > 0 and k not in W and k != 'reward_learned_embed']
    assert not missing, f'fixture lacks keys {missing}'
    with torch.no_grad():
        for k, p in own.items():
            if k in W:
                assert tuple(p.shape) == tuple(W[k].shape), (k, p.shape, W[k].shape)
                p.copy_(W[k])
    return m


def t(a):
    return torch.from_numpy(np.asarray(a))


def golden_noise(g, prefix):
    return {k: t(g[prefix + 'noise_' + k]) for k in ('latent', 'context', 'gumbel_u', 'bern_u', 'beta') if prefix + 'noise_' + k in g}


def rollout_parity(e, ref, nz, cfg, margin=1e-3):
    """A product rollout `e` (Experience) against the oracle's `ref` (restate.generate) under the same injected noise `nz`, at any batch size.
    Exact-index parity is well posed only where the oracle's own decision margins exceed fp32 noise: per trajectory, the smallest
    top-2 gap of (logit + Gumbel) over its sampled actions is recovered from the oracle's outputs; a trajectory below `margin` is 'ill posed' and
    reported, not compared (an index that flips there sends the whole trajectory down
    another path).  Returns a dict of plain numbers: counts, integer equality on the well-posed trajectories, max |diff| of the float fields there."""
    F_ = ref['latents'].shape[1]
    B = ref['latents'].shape[0]
    well = torch.ones(B, dtype=torch.bool)
    lg = ref.get('old_action_unembeds')
    if lg is not None and lg.numel() > 0:
        Fa = lg.shape[1]
        u = nz['gumbel_u'][:Fa].transpose(0, 1)[:B]
        g = -torch.log((-torch.log(u.clamp(min=1e-20))).clamp(min=1e-20))
        z, o = lg + g, 0
        for n in cfg.num_discrete_actions:
            top = z[..., o:o + n].topk(2, dim=-1).values
            gap = top[..., 0] - top[..., 1]                                            # (B, Fa)
            valid = torch.arange(Fa)[None] < ref['lens'][:, None]
            well &= (gap.masked_fill(~valid, float('inf')).min(dim=1).values >= margin)
            o += n
    out = dict(trajectories=B, frames=F_, well_posed_trajectories=int(well.sum()), margin=margin)
    same_len = e.latents.shape[1] == F_
    out['frames_equal'] = bool(same_len)
    if not same_len or not well.any():
        return out
    c = lambda x: x.detach().cpu()
    acts = c(e.actions.discrete) if e.actions is not None and e.actions.discrete is not None else None
    if acts is not None and ref.get('actions') is not None:
        eq = (acts == ref['actions']).flatten(1).all(dim=1)
        out['actions_equal'] = bool(eq[well].all())
        out['trajectories_with_identical_actions'] = int(eq.sum())
        well = well & eq if not out['actions_equal'] else well
        if not well.any():
            return out
    out['lens_equal'] = bool(torch.equal(c(e.lens)[well], ref['lens'][well]))
    if 'terminals' in ref and e.terminals is not None:
        out['terminals_equal'] = bool(torch.equal(c(e.terminals)[well], ref['terminals'][well]))
    for name, a, b in (('latents', e.latents, ref['latents']), ('agent_embed', e.agent_embed, ref.get('agent_embed')),
                       ('values', e.values, ref.get('values')), ('rewards', e.rewards, ref.get('rewards')),
                       ('log_probs', e.log_probs.discrete if e.log_probs is not None else None, ref.get('log_probs'))):
        if a is None or b is None:
            continue
        a, b = c(a).float()[well], b.float()[well]
        out[name + '_max_abs'] = float((a - b).abs().max())
        out[name + '_scale'] = float(b.abs().max())
    return out


def rollout_parity_continuous(e, ref, track_tol=5e-2):
    """A REDUCED-PRECISION product rollout `e` (the bf16 engine, continuous actions only) against the oracle's fp32 `ref` under the same injected noise.
    A Beta draw is a ratio of rejection-sampled gammas: an accept / reject decision whose margin is below the engine's error flips, the sample jumps
    and that trajectory takes another path from there on (the action feeds the next frames).  A trajectory is 'tracked' while every sampled action stays
    within `track_tol` of the oracle's; the float fields are compared on the tracked trajectories (and, separately, on all of them).  Plain numbers."""
    c = lambda x: x.detach().cpu().float()
    B, F_ = ref['latents'].shape[:2]
    out = dict(trajectories=B, frames=F_, frames_equal=bool(e.latents.shape[1] == F_), track_tol=track_tol)
    if not out['frames_equal']:
        return out
    da = (c(e.actions.continuous) - ref['actions_cont'].float()).abs().flatten(1).max(dim=1).values
    tracked = da <= track_tol
    out['tracked_trajectories'] = int(tracked.sum())
    out['actions_cont_max_abs_all'] = float(da.max())
    fields = (('latents', e.latents, ref['latents']), ('agent_embed', e.agent_embed, ref.get('agent_embed')), ('values', e.values, ref.get('values')),
              ('rewards', e.rewards, ref.get('rewards')), ('cont_logp', e.log_probs.continuous if e.log_probs is not None else None, ref.get('log_probs_cont')),
              ('actions_cont', e.actions.continuous, ref.get('actions_cont')))
    for name, a, b in fields:
        if a is None or b is None:
            continue
        d = (c(a) - b.float()).abs().flatten(1)
        out[name + '_scale'] = float(b.abs().max())
        out[name + '_max_abs_all'] = float(d.max())
        if tracked.any():
            out[name + '_max_abs'] = float(d[tracked].max())
            out[name + '_mean_abs'] = float(d[tracked].mean())
    return out


def first_trajectories(exp, b):
    """The first `b` trajectories of an Experience: every tensor whose leading dimension is the batch (the payload's) is sliced, anything else —
    scalars, per-model tensors of another leading size — is kept as it is."""
    from dataclasses import fields, replace
    from torch.utils._pytree import tree_map
    payload = exp.latents if exp.latents is not None else exp.video
    batch = payload.shape[0]
    cut = lambda v: v[:b] if torch.is_tensor(v) and v.ndim >= 1 and v.shape[0] == batch else v
    return replace(exp, **{f.name: tree_map(cut, getattr(exp, f.name)) for f in fields(exp)})
