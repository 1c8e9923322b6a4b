"""Generate tests/golden/*.npz from the REFERENCE itself (build container only).

    python -m oracle.gen_golden          # needs /root/reference; writes tests/golden/

The reference's dreamer4/dreamer4.py is imported unmodified through oracle/shim (stand-ins for the
third-party packages this image lacks, see oracle/ref_import.py).  Each fixture records
`oracle = "shim"` and which third-party behaviours were restated rather than imported.  Every random
draw of DynamicsWorldModel.generate is injected (oracle/ref_harness.py) so the fixtures are pure
functions of the stored inputs.

Fixtures (model: dim 32, 2 heads x 64, 6 latent tokens x 8, depth 4, time block every 2):
  weights.npz      reference state_dict of the fixture model
  generate.npz     generate() in four modes: cached, no time cache, 2-frame prompt, 3 chained calls
  forward.npz      one parallel forward over 4 frames (+ the same frames fed one at a time with the cache)
  variant.npz      a second architecture (weights_variant.npz): generate cached / uncached, ppo + pmpo losses and gradients
  samelen.npz      num_spatial_tokens == num_latent_tokens (weights_samelen.npz): rollout + env-wrapper style chained calls
  headdim16.npz    attn_dim_head = 16 (weights_headdim16.npz): rollout, and a rollout returning the time KV cache
  actionfree.npz   a world model without an action space (weights_actionfree.npz): plain and rewards-only rollouts
  options.npz      non-default call options on the main model (context noise, temperatures, 64 denoising steps, store_* = False)
  hyper.npz        non-default hyper-parameters (weights_hyper.npz): head depths, value / reward ranges and bins, max_steps, softclamp,
                   GAE / PPO / PMPO / entropy constants: rollout + ppo / spo / pmpo losses and gradients
  noterm.npz       predict_terminals = False (weights_noterm.npz): policy-optimisation rollout with tasks, ppo losses
  blocks.npz       block-level intermediates of that parallel forward (forward hooks on the reference's modules)
  learn.npz        learn_from_experience ppo / spo / pmpo: losses + head gradients (autograd), GAE returns
  trainer.npz      3 DreamTrainer-style steps (trainers.py:1430-1452): losses, grad norms, final head weights
  postln.npz       head MLPs in the Linear -> LayerNorm -> SiLU recipe (weights_postln.npz): rollout, ppo / pmpo losses and gradients
  continuous.npz   continuous (Beta) actions: a mixed discrete + continuous model (weights_continuous.npz: rollout, ppo / spo / pmpo
                   losses and gradients) and a continuous-only one (weights_contonly.npz: tempered rollout, env-wrapper chained calls)
  symexp.npz       reward_encoder_type='symexp_two_hot' (weights_symexp.npz): rollout, ppo losses and gradients
  encode.npz       VideoTokenizer.tokenize of reference tokenizers (weights_encode*.npz = the encoder half) + generate(prompt=video)
  train.npz        flow + shortcut losses of the dynamics training forward and their gradients (weights_train.npz)
  train_agent.npz  the whole training forward with rewards / terminals / actions (weights_train_agent.npz)
  train_cont.npz   the training forward with discrete + continuous action cloning (weights_train_cont.npz)
  decode.npz       VideoTokenizer.decode of a reference tokenizer (weights_decode.npz = the decoder half of its state_dict): two flow steps
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_harness import build_reference_model, injected, make_noise, weights_of   # noqa: E402
from oracle.ref_import import load_reference, third_party_sources                       # noqa: E402
from oracle.restate import Config                                                       # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
META = dict(
    oracle='shim',
    reference='lucidrains/dreamer4 v0.16.3 dreamer4/dreamer4.py imported unmodified',
    restated_third_party='x_mlps_pytorch(create_mlp, Ensemble) hl_gauss_pytorch discrete_continuous_embed_readout(MultiCategorical, Readout, BetaDist) '
                         'assoc_scan einx torch_einops_utils  -- PARITY UNPINNED against the real packages',
)

CFG = dict(dim=32, dim_latent=8, num_latent_tokens=6, depth=4, time_block_every=2, attn_heads=2, attn_dim_head=64,
           num_discrete_actions=(4,), num_tasks=3, reward_num_bins=63, value_num_bins=63, multi_token_pred_len=4)


CFG_VARIANT = dict(dim=48, dim_latent=4, num_latent_tokens=5, depth=3, time_block_every=1, attn_heads=1, attn_dim_head=64,
                   num_spatial_tokens=2, num_register_tokens=0, num_discrete_actions=(3, 2), num_tasks=0, reward_num_bins=31,
                   value_num_bins=31, multi_token_pred_len=1)


CFG_SAMELEN = dict(dim=32, dim_latent=16, num_latent_tokens=4, num_spatial_tokens=4, depth=2, time_block_every=1, attn_heads=2,
                   attn_dim_head=64, num_discrete_actions=(4,), num_tasks=0, reward_num_bins=31, value_num_bins=31, multi_token_pred_len=8)


CFG_HEADDIM16 = dict(dim=32, dim_latent=8, num_latent_tokens=6, depth=3, time_block_every=2, attn_heads=3, attn_dim_head=16,
                     num_discrete_actions=(4,), num_tasks=0, reward_num_bins=31, value_num_bins=31, multi_token_pred_len=2)


CFG_ACTIONFREE = dict(dim=16, dim_latent=8, num_latent_tokens=6, depth=2, time_block_every=2, attn_heads=1, attn_dim_head=32,
                      num_discrete_actions=(), num_tasks=0, reward_num_bins=31, value_num_bins=31, multi_token_pred_len=1)


CFG_HYPER = dict(dim=16, dim_latent=8, num_latent_tokens=6, depth=2, time_block_every=2, attn_heads=1, attn_dim_head=32,
                 num_discrete_actions=(3,), num_tasks=0, max_steps=16, attn_softclamp_value=30., reward_num_bins=15, reward_range=(-5., 5.),
                 value_num_bins=21, value_range=(-10., 10.), multi_token_pred_len=2, policy_head_mlp_depth=2, value_head_mlp_depth=1,
                 gae_discount_factor=0.9, gae_lambda=0.8, ppo_eps_clip=0.1, policy_entropy_weight=0.05, use_delight_gating=False,
                 pmpo_pos_to_neg_weight=0.3, pmpo_reverse_kl=False, pmpo_kl_div_loss_weight=0.5)


CFG_NOTERM = dict(dim=16, dim_latent=4, num_latent_tokens=3, depth=1, time_block_every=1, attn_heads=2, attn_dim_head=16,
                  num_discrete_actions=(5,), num_tasks=2, predict_terminals=False, reward_num_bins=15, value_num_bins=15, multi_token_pred_len=1)


CFG_POSTLN = dict(dim=32, dim_latent=8, num_latent_tokens=4, depth=2, time_block_every=2, attn_heads=2, attn_dim_head=32,
                  num_discrete_actions=(4,), num_tasks=0, reward_num_bins=31, value_num_bins=31, multi_token_pred_len=2,
                  policy_head_mlp_depth=2, value_head_mlp_depth=1, head_mlp_recipe='post_layer')


CFG_CONT = dict(dim=32, dim_latent=8, num_latent_tokens=4, depth=2, time_block_every=2, attn_heads=2, attn_dim_head=32,
                num_discrete_actions=(3,), num_continuous_actions=3, num_tasks=0, reward_num_bins=31, value_num_bins=31,
                multi_token_pred_len=2, policy_head_mlp_depth=1, value_head_mlp_depth=1)


CFG_CONTONLY = dict(dim=16, dim_latent=4, num_latent_tokens=3, depth=2, time_block_every=1, attn_heads=1, attn_dim_head=16,
                    num_discrete_actions=(), num_continuous_actions=2, num_tasks=0, reward_num_bins=15, value_num_bins=15,
                    multi_token_pred_len=1, policy_head_mlp_depth=1, value_head_mlp_depth=1)


CFG_BETAEXP = dict(CFG_CONTONLY, num_discrete_actions=(3,), num_continuous_actions=3, continuous_beta_param='exp_p1')


def fixture_config():
    return Config(**CFG)


def npy(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def exp_dict(prefix, e, out):
    out[prefix + 'latents'] = npy(e.latents)
    out[prefix + 'agent_embed'] = npy(e.agent_embed)
    out[prefix + 'rewards'] = npy(e.rewards)
    out[prefix + 'values'] = npy(e.values)
    if e.actions.discrete is not None:
        out[prefix + 'log_probs'] = npy(e.log_probs.discrete)
        out[prefix + 'actions'] = npy(e.actions.discrete)
        out[prefix + 'unembeds'] = npy(e.old_action_unembeds.discrete)
    if e.actions.continuous is not None:
        out[prefix + 'log_probs_cont'] = npy(e.log_probs.continuous)
        out[prefix + 'actions_cont'] = npy(e.actions.continuous)
        out[prefix + 'cont_params'] = npy(e.old_action_unembeds.continuous)
    out[prefix + 'lens'] = npy(e.lens)
    out[prefix + 'terminals'] = npy(e.terminals)
    out[prefix + 'episode_return'] = npy(e.episode_return)


def min_margin_multi(e, noise, cfg):
    """min_margin for several action types: the top-2 gap is taken inside each type's slice of the logits."""
    lg = e.old_action_unembeds.discrete
    F = lg.shape[1]
    u = noise['gumbel_u'][:F].transpose(0, 1)
    g = -torch.log((-torch.log(u.clamp(min=1e-20))).clamp(min=1e-20))
    z, o, best = lg + g, 0, float('inf')
    for n in cfg.num_discrete_actions:
        top = z[..., o:o + n].topk(2, dim=-1).values
        best = min(best, float((top[..., 0] - top[..., 1]).min()))
        o += n
    return best


def noise_dict(prefix, nz, out):
    for k, v in nz.items():
        out[prefix + 'noise_' + k] = npy(v)


def min_margin(e, noise, cfg):
    """Smallest top-2 gap of (logit + gumbel) over all sampled actions: exact-index parity is only
    well posed when this is comfortably above the fp32 tolerance."""
    lg = e.old_action_unembeds.discrete                    # (B, F, A)
    F = lg.shape[1]
    u = noise['gumbel_u'][:F].transpose(0, 1)
    g = -torch.log((-torch.log(u.clamp(min=1e-20))).clamp(min=1e-20))
    top = (lg + g).topk(2, dim=-1).values
    return float((top[..., 0] - top[..., 1]).min())


HEADS = ('policy_head', 'value_head', 'action_embedder.discrete_action_unembed', 'action_embedder.continuous_action_unembed')


def save_weights(name, W, cfgd):
    np.savez(os.path.join(OUT, name), **{k: npy(v) for k, v in W.items() if v.numel() > 0},
             **{'meta_' + k: np.array(v) for k, v in META.items()}, **{'cfg_' + k: np.array(v) for k, v in cfgd.items()})


def learn_into(out, m, e, objectives):
    for obj in objectives:
        m.zero_grad()
        pl_, vl_ = m.learn_from_experience(e, objective=obj)
        pl_.backward(); vl_.backward()
        out[f'{obj}_policy_loss'], out[f'{obj}_value_loss'] = npy(pl_), npy(vl_)
        for k, p in m.named_parameters():
            if k.startswith(HEADS) and p.numel() > 0 and p.grad is not None:
                if p.ndim == 1 or 'unembed' in k or p.numel() <= 4096:
                    out[f'{obj}_grad/{k}'] = npy(p.grad)
                else:
                    out[f'{obj}_gnorm/{k}'] = npy(p.grad.norm())
                    out[f'{obj}_gsample/{k}'] = npy(p.grad.flatten()[::97])


def gen_postln():
    """postln.npz / weights_postln.npz: the head MLPs in the OTHER plausible recipe of x_mlps_pytorch's normed MLP
    (Linear -> LayerNorm -> SiLU, bare last Linear; oracle/shim/x_mlps_pytorch/normed_mlp.py RECIPE = 'post_layer')."""
    cfg = Config(**CFG_POSTLN)
    m = build_reference_model(cfg, seed=21)
    with torch.no_grad():
        m.action_embedder.discrete_action_unembed.mul_(0.3)
        g = torch.Generator().manual_seed(22)
        for k, p in m.named_parameters():          # LayerNorm biases default to zero: make them visible
            if k.startswith(HEADS + ('to_state_terminal_pred',)) and k.endswith('.1.bias') and p.ndim == 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    W = weights_of(m)
    assert 'policy_head.layers.0.1.bias' in W and f'policy_head.layers.{cfg.policy_head_mlp_depth + 1}.weight' in W
    save_weights('weights_postln.npz', W, CFG_POSTLN)
    out = {}
    nz = make_noise(cfg, 5, 3, 901)
    with injected(nz):
        e = m.generate(5, batch_size=3, return_for_policy_optimization=True)
    exp_dict('cached_', e, out); noise_dict('cached_', nz, out)
    out['cached_margin'] = np.array(min_margin(e, nz, cfg))
    learn_into(out, m, e, ('ppo', 'pmpo'))
    # an experience generated with store_agent_embed=False: learn_from_experience re-runs the world model over the stored latents
    # to get the agent embeddings (D4:6045-6070)
    nz = make_noise(cfg, 4, 3, 902)
    with injected(nz):
        e2 = m.generate(4, batch_size=3, return_for_policy_optimization=True, store_agent_embed=False, return_terminals=False)
    assert e2.agent_embed is None
    exp_dict_noembed = dict(latents=e2.latents, rewards=e2.rewards, values=e2.values, log_probs=e2.log_probs.discrete, actions=e2.actions.discrete,
                            lens=e2.lens, terminals=e2.terminals, unembeds=e2.old_action_unembeds.discrete)
    for k, v in exp_dict_noembed.items():
        out['noembed_' + k] = npy(v)
    m.zero_grad()
    pl_, vl_ = m.learn_from_experience(e2, objective='ppo')
    pl_.backward(); vl_.backward()
    out['noembed_ppo_policy_loss'], out['noembed_ppo_value_loss'] = npy(pl_), npy(vl_)
    out['noembed_ppo_grad_unembed'] = npy(m.action_embedder.discrete_action_unembed.grad)
    np.savez(os.path.join(OUT, 'postln.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    print('postln margin', out['cached_margin'], 'lens', out['cached_lens'])


def gen_continuous():
    """continuous.npz: continuous (Beta) actions through the Readout / BetaDist stand-in — a mixed discrete + continuous model
    (rollout, ppo / spo / pmpo losses and gradients) and a continuous-only one (tempered sampling, env-wrapper style chained calls
    with prompt_continuous_actions and the carried time cache)."""
    from oracle import restate
    cfg = Config(**CFG_CONT)
    m = build_reference_model(cfg, seed=31)
    with torch.no_grad():
        m.action_embedder.discrete_action_unembed.mul_(0.3)
    W = weights_of(m)
    save_weights('weights_continuous.npz', W, CFG_CONT)
    out = {}

    def beta_margin(e, nz, temperature=1.):
        F = e.old_action_unembeds.continuous.shape[1]
        return restate.beta_accept_margin(e.old_action_unembeds.continuous, nz['beta'][:F].transpose(0, 1), temperature)

    # the noise seed is the first one whose draws are decided with a comfortable margin: the accept / reject decisions of the gamma
    # sampler and the discrete arg-max are then the same in every fp32 implementation (SURVEY.md 8c "margin")
    for seed in range(911, 960):
        nz = make_noise(cfg, 5, 3, seed)
        with injected(nz):
            e = m.generate(5, batch_size=3, return_for_policy_optimization=True)
        if beta_margin(e, nz) >= 2e-3 and min_margin(e, nz, cfg) >= 1e-2 and int(e.lens.min()) >= 2:
            break
    exp_dict('cached_', e, out); noise_dict('cached_', nz, out)
    out['cached_seed'] = np.array(seed)
    out['cached_margin'] = np.array(min_margin(e, nz, cfg))
    out['cached_beta_margin'] = np.array(beta_margin(e, nz))
    learn_into(out, m, e, ('ppo', 'spo', 'pmpo'))
    # continuous-only model
    cfg2 = Config(**CFG_CONTONLY)
    m2 = build_reference_model(cfg2, seed=33)
    W2 = weights_of(m2)
    save_weights('weights_contonly.npz', W2, CFG_CONTONLY)
    for seed in range(921, 960):
        nz = make_noise(cfg2, 4, 2, seed)
        with injected(nz):
            e = m2.generate(4, batch_size=2, return_for_policy_optimization=True, continuous_temperature=0.7)
        if beta_margin(e, nz, 0.7) >= 2e-3 and int(e.lens.min()) >= 2:
            break
    exp_dict('only_', e, out); noise_dict('only_', nz, out)
    out['only_beta_margin'] = np.array(beta_margin(e, nz, 0.7))
    m2.zero_grad()
    pl_, vl_ = m2.learn_from_experience(e, objective='ppo')
    out['only_ppo_policy_loss'], out['only_ppo_value_loss'] = npy(pl_), npy(vl_)
    for seed in range(931, 980):
        nz = make_noise(cfg2, 3, 2, seed)
        lat_hist = torch.zeros(2, 0, 3, 4); act_hist = torch.zeros(2, 0, 2); tc = None
        env, worst = {}, float('inf')
        for i in range(3):
            sub = {k: v[i:i + 1] for k, v in nz.items()}
            kw = dict(prompt_latents=lat_hist, prompt_continuous_actions=act_hist) if i > 0 else {}
            with injected(sub):
                e, tc = m2.generate(i + 1, batch_size=2, return_rewards_per_frame=True, return_agent_actions=True,
                                    return_log_probs_and_values=True, time_cache=tc, return_time_cache=True, return_terminals=False, **kw)
            worst = min(worst, beta_margin(e, sub))
            lat_hist, act_hist = e.latents, e.actions.continuous
            env[f'env{i}_latents'], env[f'env{i}_actions_cont'], env[f'env{i}_values'] = npy(e.latents), npy(e.actions.continuous), npy(e.values)
        if worst >= 2e-3:
            break
    out.update(env); noise_dict('env_', nz, out)
    out['env_beta_margin'] = np.array(worst)
    np.savez(os.path.join(OUT, 'continuous.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    print('continuous margins', out['cached_margin'], out['cached_beta_margin'], out['only_beta_margin'], out['env_beta_margin'], 'lens', out['cached_lens'], out['only_lens'])


def gen_beta_exp():
    """beta_exp.npz / weights_beta_exp.npz: the Beta head with the OTHER link of its raw parameters (alpha = exp(raw) + 1; stand-in switch
    BETA_PARAM = 'exp_p1'): rollout with tempered continuous sampling, ppo / pmpo losses and head gradients (entropy bonus and the
    Beta KL of pmpo are the terms whose derivative changes with the link)."""
    from oracle import restate
    cfg = Config(**CFG_BETAEXP)
    m = build_reference_model(cfg, seed=37)
    with torch.no_grad():
        m.action_embedder.discrete_action_unembed.mul_(0.3)
    W = weights_of(m)
    save_weights('weights_beta_exp.npz', W, CFG_BETAEXP)
    out = {}

    def beta_margin(e, nz, temperature):
        F = e.old_action_unembeds.continuous.shape[1]
        return restate.beta_accept_margin(e.old_action_unembeds.continuous, nz['beta'][:F].transpose(0, 1), temperature, 'exp_p1')

    for seed in range(941, 990):
        nz = make_noise(cfg, 5, 3, seed)
        with injected(nz):
            e = m.generate(5, batch_size=3, return_for_policy_optimization=True, continuous_temperature=0.8)
        if beta_margin(e, nz, 0.8) >= 2e-3 and min_margin(e, nz, cfg) >= 1e-2 and int(e.lens.min()) >= 2:
            break
    exp_dict('cached_', e, out); noise_dict('cached_', nz, out)
    out['cached_seed'] = np.array(seed)
    out['cached_margin'] = np.array(min_margin(e, nz, cfg))
    out['cached_beta_margin'] = np.array(beta_margin(e, nz, 0.8))
    learn_into(out, m, e, ('ppo', 'pmpo'))
    np.savez(os.path.join(OUT, 'beta_exp.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    a, b = restate.beta_alpha_beta(e.old_action_unembeds.continuous, 'exp_p1')
    print('beta_exp margins', out['cached_margin'], out['cached_beta_margin'], 'lens', out['cached_lens'], 'alpha range', float(a.min()), float(a.max()),
          'ppo', out['ppo_policy_loss'], 'pmpo', out['pmpo_policy_loss'])


CFG_DECODE = dict(dim=32, dim_latent=8, patch_size=4, image_height=16, image_width=24, num_latent_tokens=6, decoder_depth=3, time_block_every=2,
                  attn_heads=2, attn_dim_head=64, channels=3, decoder_pos_mlp_depth=2, decoder_flow_steps=2)


def gen_decode():
    """decode.npz / weights_decode.npz: VideoTokenizer.decode (D4:4186-4237) of the reference tokenizer — the decoder half of its
    state_dict, latents, the injected initial flow sample (D4:4212) and the decoded video; two flow steps, one time layer."""
    D4 = load_reference()
    torch.manual_seed(41)
    tok = D4.VideoTokenizer(dim=CFG_DECODE['dim'], dim_latent=CFG_DECODE['dim_latent'], patch_size=CFG_DECODE['patch_size'],
                            image_height=CFG_DECODE['image_height'], image_width=CFG_DECODE['image_width'],
                            num_latent_tokens=CFG_DECODE['num_latent_tokens'], encoder_depth=1, decoder_depth=CFG_DECODE['decoder_depth'],
                            time_block_every=CFG_DECODE['time_block_every'], attn_heads=CFG_DECODE['attn_heads'], attn_dim_head=CFG_DECODE['attn_dim_head'],
                            channels=CFG_DECODE['channels'], decoder_pos_mlp_depth=CFG_DECODE['decoder_pos_mlp_depth'],
                            decoder_flow_steps=CFG_DECODE['decoder_flow_steps'], lpips_loss_weight=0.).eval()
    g = torch.Generator().manual_seed(42)
    with torch.no_grad():
        for k, p in tok.named_parameters():
            if p.ndim == 1 and ('norm' in k or k.endswith('.0.weight')):
                p.copy_(1. + torch.randn(p.shape, generator=g) * 0.1)
            if k.endswith('gamma'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    keep = ('latents_to_decoder.', 'time_embed.', 'noised_patch_to_tokens.', 'decoder.')
    W = {k: v.detach().clone().float() for k, v in tok.state_dict().items() if k.startswith(keep)}
    save_weights('weights_decode.npz', W, CFG_DECODE)
    B, T = 2, 3
    lat = torch.randn(B, T, 6, 8, generator=g).clamp(-1., 1.)
    noise = torch.randn(B, 3, T, 16, 24, generator=g)
    saved = D4.randn
    D4.randn = lambda *a, **k: noise.clone()
    try:
        with torch.no_grad():
            video, preds = tok.decode(lat, return_recons_across_steps=True)
    finally:
        D4.randn = saved
    out = dict(latents=npy(lat), noise=npy(noise), video=npy(video), pred_step0=npy(preds[0]))
    np.savez(os.path.join(OUT, 'decode.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    print('decode video', tuple(video.shape), 'abs max', float(video.abs().max()))


CFG_ENCODE = dict(dim=32, dim_latent=8, patch_size=4, image_height=16, image_width=24, num_latent_tokens=6, encoder_depth=3, decoder_depth=1,
                  time_block_every=2, attn_heads=2, attn_dim_head=64, channels=3)
CFG_ENCODE_WIDE = dict(dim=32, dim_latent=4, patch_size=4, image_height=32, image_width=40, num_latent_tokens=5, encoder_depth=2, decoder_depth=1,
                       time_block_every=2, attn_heads=1, attn_dim_head=64, channels=1)
CFG_ENCODE_DYN = dict(dim=32, dim_latent=8, num_latent_tokens=6, depth=2, time_block_every=2, attn_heads=2, attn_dim_head=32,
                      num_discrete_actions=(4,), num_tasks=0, reward_num_bins=11, value_num_bins=11, reward_range=(-3., 3.), value_range=(-4., 4.),
                      multi_token_pred_len=2, policy_head_mlp_depth=1, value_head_mlp_depth=1)


def _reference_tokenizer(D4, cfgd, seed):
    torch.manual_seed(seed)
    tok = D4.VideoTokenizer(**cfgd, lpips_loss_weight=0.).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, p in tok.named_parameters():
            if p.ndim == 1 and ('norm' in k or k.endswith('.0.weight')):
                p.copy_(1. + torch.randn(p.shape, generator=g) * 0.1)
            if k.endswith('gamma'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        tok.latent_tokens.mul_(30.)                     # init std 1e-2: make the learned tokens matter
    return tok, g


ENCODER_KEYS = ('latent_tokens', 'patch_to_tokens.', 'encoder_transformer.', 'encoded_to_latents.')


def gen_encode():
    """encode.npz / weights_encode*.npz: VideoTokenizer.tokenize (D4:4107-4113 -> eval forward(return_latents=True), D4:4239-4433) of the
    reference tokenizer — the encoder half of its state_dict, a video and its latents; one tokenizer with a time layer (30 tokens per
    frame), one with 80 patches per frame (the latent tokens' cross attention over > 64 keys); and DynamicsWorldModel.generate with a
    video prompt (D4:6376-6387) on the first."""
    D4 = load_reference()
    out = {}
    tok, g = _reference_tokenizer(D4, CFG_ENCODE, 61)
    save_weights('weights_encode.npz', {k: v.detach().clone().float() for k, v in tok.state_dict().items() if k.startswith(ENCODER_KEYS)}, CFG_ENCODE)
    video = torch.rand(2, 3, 4, 16, 24, generator=g)
    with torch.no_grad():
        out['video'], out['latents'] = npy(video), npy(tok.tokenize(video))
        image = torch.rand(3, 3, 16, 24, generator=g)
        out['image'], out['image_latents'] = npy(image), npy(tok.tokenize(image))
    tokw, g = _reference_tokenizer(D4, CFG_ENCODE_WIDE, 63)
    save_weights('weights_encode_wide.npz', {k: v.detach().clone().float() for k, v in tokw.state_dict().items() if k.startswith(ENCODER_KEYS)}, CFG_ENCODE_WIDE)
    videow = torch.rand(2, 1, 3, 32, 40, generator=g)
    with torch.no_grad():
        out['wide_video'], out['wide_latents'] = npy(videow), npy(tokw.tokenize(videow))
    # a dynamics model prompted with a video: the prompt frames are tokenized, then teacher-forced (D4:6376-6395)
    cfg = Config(**CFG_ENCODE_DYN)
    m = build_reference_model(cfg, seed=65)
    with torch.no_grad():
        m.action_embedder.discrete_action_unembed.mul_(0.3)
    save_weights('weights_encode_dyn.npz', weights_of(m), CFG_ENCODE_DYN)
    m.video_tokenizer = tok
    nz = make_noise(cfg, 5, 2, 967)
    pa = torch.randint(0, 4, (2, 2, 1), generator=g)
    pr = torch.randn(2, 2, generator=g)
    with injected(nz):
        e = m.generate(5, batch_size=2, prompt=video[:, :, :2], prompt_discrete_actions=pa, prompt_rewards=pr,
                       return_for_policy_optimization=True, return_decoded_video=False)
    exp_dict('prompt_', e, out); noise_dict('prompt_', nz, out)
    out['prompt_actions_in'], out['prompt_rewards_in'] = npy(pa), npy(pr)
    out['prompt_margin'] = np.array(min_margin(e, nz, cfg))
    np.savez(os.path.join(OUT, 'encode.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    print('encode latents', out['latents'].shape, 'abs max', float(np.abs(out['latents']).max()), 'std', float(out['latents'].std()),
          'wide', out['wide_latents'].shape, float(out['wide_latents'].std()), 'prompt margin', out['prompt_margin'], 'lens', out['prompt_lens'])


CFG_TRAIN = dict(dim=32, dim_latent=8, num_latent_tokens=6, depth=3, time_block_every=2, attn_heads=2, attn_dim_head=32,
                 num_discrete_actions=(4,), num_tasks=0, reward_num_bins=11, value_num_bins=11, multi_token_pred_len=1, max_steps=16)


def gen_train():
    """train.npz / weights_train.npz: the flow and shortcut losses of DynamicsWorldModel.forward in training (D4:6956-7003, 7335-7431)
    and the gradient of (flow + shortcut) with respect to every parameter on that path; the model's own random draws (shortcut coin,
    step sizes, signal levels, noise — D4:6965-6977, 7000) are recorded as they are made, under `seed=`.  One shortcut batch, one plain
    flow batch."""
    D4 = load_reference()
    cfg = Config(**CFG_TRAIN)
    m = build_reference_model(cfg, seed=71)
    W = weights_of(m)
    save_weights('weights_train.npz', W, CFG_TRAIN)
    g = torch.Generator().manual_seed(72)
    B, T = 3, 4
    lat = torch.randn(B, T, 6, 8, generator=g).clamp(-2, 2)
    acts = torch.randint(0, 4, (B, T, 1), generator=g)
    out = dict(latents=npy(lat), actions=npy(acts))
    for name, prob in (('shortcut', 1.), ('plain', 0.)):
        rec = {}
        saved = (D4.randint, D4.randn_like, D4.sample_prob)

        def rec_randint(*a, **k):
            r = saved[0](*a, **k); rec.setdefault('randint', []).append(r.clone()); return r

        def rec_randn_like(*a, **k):
            r = saved[1](*a, **k); rec.setdefault('randn_like', []).append(r.clone()); return r

        D4.randint, D4.randn_like = rec_randint, rec_randn_like
        m.prob_shortcut_train = prob
        try:
            m.zero_grad()
            total, losses = m(latents=lat, discrete_actions=acts, seed=5, return_all_losses=True, add_autoregressive_action_loss=False)
        finally:
            D4.randint, D4.randn_like, D4.sample_prob = saved
        (losses.flow + losses.shortcut).backward()
        if prob == 1.:
            step_log2, sig_raw = rec['randint']
            sig = sig_raw // (2 ** step_log2)[:, None] * (2 ** step_log2)[:, None]
        else:
            step_log2, sig = torch.zeros(B, dtype=torch.long), rec['randint'][0]
        noise = rec['randn_like'][0][:, :, 0]                  # (b t 1 n d) -> (b t n d)
        out[f'{name}_step_sizes_log2'], out[f'{name}_signal_levels'], out[f'{name}_noise'] = npy(step_log2), npy(sig), npy(noise)
        out[f'{name}_flow_loss'], out[f'{name}_shortcut_loss'] = npy(losses.flow), npy(losses.shortcut)
        ng = 0
        for k, p in m.named_parameters():
            if p.grad is not None and p.numel() > 0 and float(p.grad.abs().max()) > 0:
                out[f'{name}_grad/{k}'] = npy(p.grad); ng += 1
        print(name, 'flow', float(losses.flow), 'shortcut', float(losses.shortcut), 'step_log2', step_log2.tolist(), 'grads', ng)
    np.savez(os.path.join(OUT, 'train.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})


CFG_TRAIN_AGENT = dict(dim=32, dim_latent=8, num_latent_tokens=6, depth=3, time_block_every=2, attn_heads=2, attn_dim_head=32,
                       num_discrete_actions=(4, 3), num_tasks=0, reward_num_bins=11, value_num_bins=11, multi_token_pred_len=2, max_steps=16)


def gen_train_agent():
    """train_agent.npz / weights_train_agent.npz: the whole training forward (D4:6956-7743) with rewards, terminals and two discrete
    action types, multi-token prediction length 2: flow / rewards / terminals / discrete-action losses, the total, and its gradient
    with respect to every parameter; draws recorded as in `train`."""
    D4 = load_reference()
    cfg = Config(**CFG_TRAIN_AGENT)
    m = build_reference_model(cfg, seed=73, head_scale=False)
    with torch.no_grad():
        m.action_embedder.discrete_action_unembed.mul_(10.)
        for p in m.to_reward_pred.parameters():
            if p.ndim == 3: p.mul_(3.)
    save_weights('weights_train_agent.npz', weights_of(m), CFG_TRAIN_AGENT)
    g = torch.Generator().manual_seed(74)
    B, T = 3, 5
    lat = torch.randn(B, T, 6, 8, generator=g).clamp(-2, 2)
    acts = torch.stack([torch.randint(0, 4, (B, T), generator=g), torch.randint(0, 3, (B, T), generator=g)], -1)
    rew = torch.randn(B, T, generator=g) * 2.
    term = torch.rand(B, T, generator=g) < 0.3
    out = dict(latents=npy(lat), actions=npy(acts), rewards=npy(rew), terminals=npy(term))
    rec = {}
    saved = (D4.randint, D4.randn_like)

    def rec_randint(*a, **k):
        r = saved[0](*a, **k); rec.setdefault('randint', []).append(r.clone()); return r

    def rec_randn_like(*a, **k):
        r = saved[1](*a, **k); rec.setdefault('randn_like', []).append(r.clone()); return r

    D4.randint, D4.randn_like = rec_randint, rec_randn_like
    m.prob_shortcut_train = 1.
    try:
        m.zero_grad()
        total, losses = m(latents=lat, discrete_actions=acts, rewards=rew, terminals=term, seed=9, return_all_losses=True)
    finally:
        D4.randint, D4.randn_like = saved
    total.backward()
    step_log2, sig_raw = rec['randint']
    sig = sig_raw // (2 ** step_log2)[:, None] * (2 ** step_log2)[:, None]
    out.update(step_sizes_log2=npy(step_log2), signal_levels=npy(sig), noise=npy(rec['randn_like'][0][:, :, 0]))
    out.update(total=npy(total), flow_loss=npy(losses.flow), shortcut_loss=npy(losses.shortcut), rewards_loss=npy(losses.rewards),
               terminals_loss=npy(losses.terminals), discrete_actions_loss=npy(losses.discrete_actions))
    ng = 0
    for k, p in m.named_parameters():
        if p.grad is not None and p.numel() > 0 and float(p.grad.abs().max()) > 0:
            out[f'grad/{k}'] = npy(p.grad); ng += 1
    # variable lengths (D4:7418-7426, 7460, 7488, 7586): frames past `lens` drop out of every loss term
    lens = torch.tensor([5, 3, 4])
    t_, l_ = m(latents=lat, discrete_actions=acts, rewards=rew, terminals=term, lens=lens, seed=9, return_all_losses=True)
    out['lens'] = npy(lens); out['lens_total'] = npy(t_)
    out['lens_terms'] = npy(torch.cat([l_.flow.reshape(1), l_.shortcut.reshape(1), l_.rewards, l_.terminals.reshape(1), l_.discrete_actions]))
    # the same forward with loss normalisation (D4:629-669, 5250-5255): two consecutive calls, the running mean squares evolve
    mtp = cfg.multi_token_pred_len
    m.flow_loss_normalizer, m.shortcut_flow_loss_normalizer = D4.LossNormalizer(), D4.LossNormalizer()
    m.reward_loss_normalizer, m.state_terminal_loss_normalizer = D4.LossNormalizer(mtp), D4.LossNormalizer()
    m.discrete_actions_loss_normalizer = D4.LossNormalizer(mtp)
    for call in range(2):
        t_, l_ = m(latents=lat, discrete_actions=acts, rewards=rew, terminals=term, seed=9, return_all_losses=True, update_loss_ema=True)
        out[f'norm{call}_total'] = npy(t_)
        out[f'norm{call}_terms'] = npy(torch.cat([l_.flow.reshape(1), l_.shortcut.reshape(1), l_.rewards, l_.terminals.reshape(1), l_.discrete_actions]))
    out['norm_state_flow'] = npy(m.flow_loss_normalizer.exp_avg_sq); out['norm_state_rewards'] = npy(m.reward_loss_normalizer.exp_avg_sq)
    print('train_agent total', float(total), 'flow', float(losses.flow), 'shortcut', float(losses.shortcut), 'rewards', losses.rewards.tolist(),
          'terminals', float(losses.terminals), 'actions', losses.discrete_actions.tolist(), 'grads', ng)
    np.savez(os.path.join(OUT, 'train_agent.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})


CFG_TRAIN_CONT = dict(dim=32, dim_latent=8, num_latent_tokens=6, depth=2, time_block_every=2, attn_heads=2, attn_dim_head=32,
                      num_discrete_actions=(4,), num_continuous_actions=2, num_tasks=0, reward_num_bins=11, value_num_bins=11, multi_token_pred_len=2,
                      max_steps=16)


def gen_train_cont():
    """train_cont.npz / weights_train_cont.npz: the training forward with discrete AND continuous (Beta) actions: the behaviour-cloning terms
    of both kinds at multi-token-prediction length 2 (D4:7514-7597), total and gradients.  The Beta parameterisation is the stand-in's."""
    D4 = load_reference()
    cfg = Config(**CFG_TRAIN_CONT)
    m = build_reference_model(cfg, seed=77, head_scale=False)
    with torch.no_grad():
        m.action_embedder.discrete_action_unembed.mul_(10.)
        m.action_embedder.continuous_action_unembed.mul_(10.)
    save_weights('weights_train_cont.npz', weights_of(m), CFG_TRAIN_CONT)
    g = torch.Generator().manual_seed(78)
    B, T = 3, 4
    lat = torch.randn(B, T, 6, 8, generator=g).clamp(-2, 2)
    acts = torch.randint(0, 4, (B, T, 1), generator=g)
    cont = torch.rand(B, T, 2, generator=g)
    out = dict(latents=npy(lat), actions=npy(acts), actions_cont=npy(cont))
    rec = {}
    saved = (D4.randint, D4.randn_like)

    def rec_randint(*a, **k):
        r = saved[0](*a, **k); rec.setdefault('randint', []).append(r.clone()); return r

    def rec_randn_like(*a, **k):
        r = saved[1](*a, **k); rec.setdefault('randn_like', []).append(r.clone()); return r

    D4.randint, D4.randn_like = rec_randint, rec_randn_like
    m.prob_shortcut_train = 0.
    try:
        m.zero_grad()
        total, losses = m(latents=lat, discrete_actions=acts, continuous_actions=cont, seed=11, return_all_losses=True)
    finally:
        D4.randint, D4.randn_like = saved
    total.backward()
    out.update(step_sizes_log2=npy(torch.zeros(B, dtype=torch.long)), signal_levels=npy(rec['randint'][0]), noise=npy(rec['randn_like'][0][:, :, 0]))
    out.update(total=npy(total), flow_loss=npy(losses.flow), discrete_actions_loss=npy(losses.discrete_actions), continuous_actions_loss=npy(losses.continuous_actions))
    ng = 0
    for k, p in m.named_parameters():
        if p.grad is not None and p.numel() > 0 and float(p.grad.abs().max()) > 0:
            out[f'grad/{k}'] = npy(p.grad); ng += 1
    print('train_cont total', float(total), 'discrete', losses.discrete_actions.tolist(), 'continuous', losses.continuous_actions.tolist(), 'grads', ng)
    np.savez(os.path.join(OUT, 'train_cont.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})


CFG_SYMEXP = dict(dim=32, dim_latent=8, num_latent_tokens=4, depth=2, time_block_every=2, attn_heads=2, attn_dim_head=32,
                  num_discrete_actions=(4,), num_tasks=0, reward_num_bins=41, value_num_bins=31, reward_range=(-3., 3.), value_range=(-4., 4.),
                  multi_token_pred_len=2, policy_head_mlp_depth=1, value_head_mlp_depth=1, reward_encoder_type='symexp_two_hot')


def gen_symexp():
    """symexp.npz / weights_symexp.npz: reward_encoder_type='symexp_two_hot' (SymExpTwoHot, D4:947-1040 — the reference's own code, no
    stand-in): bins -> scalar in the rollout, two-hot targets in the value loss."""
    cfg = Config(**CFG_SYMEXP)
    m = build_reference_model(cfg, seed=51)
    with torch.no_grad():
        m.action_embedder.discrete_action_unembed.mul_(0.3)
    W = weights_of(m)
    assert 'reward_encoder.bin_values' in W and 'value_encoder.bin_values' in W
    save_weights('weights_symexp.npz', W, CFG_SYMEXP)
    out = {}
    nz = make_noise(cfg, 5, 3, 951)
    with injected(nz):
        e = m.generate(5, batch_size=3, return_for_policy_optimization=True)
    exp_dict('cached_', e, out); noise_dict('cached_', nz, out)
    out['cached_margin'] = np.array(min_margin(e, nz, cfg))
    learn_into(out, m, e, ('ppo',))
    np.savez(os.path.join(OUT, 'symexp.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    print('symexp margin', out['cached_margin'], 'lens', out['cached_lens'], 'values', out['cached_values'][0])


def gen_learn_full():
    """learn_full.npz / weights_learn_full.npz: learn_from_experience(only_learn_policy_value_heads=False) (D4:6045-6075): the agent
    embeddings are recomputed by a forward WITH gradient over the stored latents at the clean signal level, so both losses reach the
    whole world model.  Losses + the gradient of each loss with respect to every parameter it reaches (trunk, embeddings, heads)."""
    cfg = Config(**CFG_TRAIN)
    m = build_reference_model(cfg, seed=81)
    with torch.no_grad():
        m.action_embedder.discrete_action_unembed.mul_(0.3)
    W = weights_of(m)
    save_weights('weights_learn_full.npz', W, CFG_TRAIN)
    out = {}
    nz = make_noise(cfg, 5, 3, 811)
    with injected(nz):
        e = m.generate(5, batch_size=3, return_for_policy_optimization=True)
    exp_dict('exp_', e, out); noise_dict('exp_', nz, out)
    out['exp_margin'] = np.array(min_margin(e, nz, cfg))
    for obj in ('ppo', 'pmpo'):
        m.zero_grad()
        pl_, vl_ = m.learn_from_experience(e, objective=obj, only_learn_policy_value_heads=False)
        out[f'{obj}_policy_loss'], out[f'{obj}_value_loss'] = npy(pl_), npy(vl_)
        pl_.backward(retain_graph=True)
        np_ = 0
        for k, p_ in m.named_parameters():
            if p_.grad is not None and p_.numel() > 0 and float(p_.grad.abs().max()) > 0:
                out[f'{obj}_pgrad/{k}'] = npy(p_.grad) if obj == 'ppo' else npy(p_.grad.norm()); np_ += 1       # pmpo: norms only (fixture size)
        m.zero_grad()
        vl_.backward()
        nv_ = 0
        for k, p_ in m.named_parameters():
            if p_.grad is not None and p_.numel() > 0 and float(p_.grad.abs().max()) > 0:
                if obj == 'ppo':
                    out[f'{obj}_vgrad/{k}'] = npy(p_.grad)
                nv_ += 1
        print(obj, 'policy', float(pl_), 'value', float(vl_), 'grads', np_, nv_)
    np.savez(os.path.join(OUT, 'learn_full.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    print('learn_full margin', out['exp_margin'], 'lens', out['exp_lens'])


EXTRA = dict(learn_full=gen_learn_full, postln=gen_postln, continuous=gen_continuous, beta_exp=gen_beta_exp, decode=gen_decode, symexp=gen_symexp, encode=gen_encode, train=gen_train, train_agent=gen_train_agent, train_cont=gen_train_cont)


def _record_third_party_sources():
    """META['oracle'] stays 'shim' while every stood-in package came from oracle/shim; under D4_ORACLE_REAL=1 with real packages
    installed it names, per package, which ones the reference imported for real (the one-flag switch)."""
    load_reference()
    src = third_party_sources()
    real = sorted(k for k, v in src.items() if v == 'real')
    if real:
        META['oracle'] = 'real: ' + ' '.join(real) + ' | shim: ' + ' '.join(sorted(k for k, v in src.items() if v != 'real'))


def main():
    _record_third_party_sources()
    if len(sys.argv) > 1:                     # python -m oracle.gen_golden postln continuous : only these families
        os.makedirs(OUT, exist_ok=True)
        for name in sys.argv[1:]:
            EXTRA[name]()
        return
    os.makedirs(OUT, exist_ok=True)
    D4 = load_reference()
    cfg = fixture_config()
    m = build_reference_model(cfg, seed=0)
    with torch.no_grad():
        m.action_embedder.discrete_action_unembed.mul_(0.3)      # 100x -> 30x: keep the policy stochastic
    W = weights_of(m)
    np.savez(os.path.join(OUT, 'weights.npz'), **{k: npy(v) for k, v in W.items() if v.numel() > 0},
             **{'meta_' + k: np.array(v) for k, v in META.items()},
             **{'cfg_' + k: np.array(v) for k, v in CFG.items()})

    # ------------------------------------------------------------------ generate
    B, T = 3, 5
    out = {}
    tasks = torch.tensor([0, 2, 1])
    nz = make_noise(cfg, T, B, 101)
    with injected(nz):
        e = m.generate(T, batch_size=B, return_for_policy_optimization=True, tasks=tasks)
    exp_dict('cached_', e, out); noise_dict('cached_', nz, out)
    out['cached_tasks'] = npy(tasks)
    out['cached_margin'] = np.array(min_margin(e, nz, cfg))
    cached_exp = e

    nz = make_noise(cfg, T, B, 102)
    with injected(nz):
        e = m.generate(T, batch_size=B, return_for_policy_optimization=True, use_time_cache=False, num_steps=2)
    exp_dict('nocache_', e, out); noise_dict('nocache_', nz, out)
    out['nocache_margin'] = np.array(min_margin(e, nz, cfg))

    g = torch.Generator().manual_seed(5)
    pl = torch.randn(B, 2, 6, 8, generator=g).clamp(-1, 1)
    pa = torch.randint(0, 4, (B, 2, 1), generator=g)
    pr = torch.randn(B, 2, generator=g)
    nz = make_noise(cfg, T, B, 103)
    with injected(nz):
        e = m.generate(T, batch_size=B, return_for_policy_optimization=True, prompt_latents=pl,
                       prompt_discrete_actions=pa, prompt_rewards=pr)
    exp_dict('prompt_', e, out); noise_dict('prompt_', nz, out)
    out['prompt_latents_in'], out['prompt_actions_in'], out['prompt_rewards_in'] = npy(pl), npy(pa), npy(pr)
    out['prompt_margin'] = np.array(min_margin(e, nz, cfg))

    nz = make_noise(cfg, 3, B, 104)
    with injected(nz):
        tc = None
        for i in range(3):
            e, tc = m.generate(1, batch_size=B, return_for_policy_optimization=True, time_cache=tc,
                               return_time_cache=True, return_terminals=False)
            exp_dict(f'chain{i}_', e, out)
    noise_dict('chain_', nz, out)
    out['chain_final_kv'] = npy(tc.main.next_kv_cache)
    # modes without action sampling (D4:6723-6729): plain latents, and rewards-only
    nz = make_noise(cfg, 4, B, 105)
    with injected(nz):
        lat = m.generate(4, batch_size=B)
    out['plain_latents'] = npy(lat); noise_dict('plain_', nz, out)
    nz = make_noise(cfg, 4, B, 106)
    with injected(nz):
        e = m.generate(4, batch_size=B, return_rewards_per_frame=True, return_terminals=True)
    out['rewonly_latents'], out['rewonly_rewards'], out['rewonly_agent_embed'] = npy(e.latents), npy(e.rewards), npy(e.agent_embed)
    out['rewonly_lens'], out['rewonly_terminals'] = npy(e.lens), npy(e.terminals)
    assert e.actions is None and e.values is None
    noise_dict('rewonly_', nz, out)

    # non-default call options on the same model -> options.npz
    opt = {}
    nz = make_noise(cfg, 4, B, 107)
    with injected(nz):
        e = m.generate(4, batch_size=B, return_for_policy_optimization=True, use_time_cache=False, context_signal_noise=0.35,
                       discrete_temperature=0.6, num_steps=8)
    exp_dict('ctxnoise_', e, opt); noise_dict('ctxnoise_', nz, opt)
    nz = make_noise(cfg, 3, B, 108)
    with injected(nz):
        e = m.generate(3, batch_size=B, return_for_policy_optimization=True, discrete_temperature=1.7, num_steps=64,
                       store_agent_embed=False, store_old_action_unembeds=False)
    assert e.agent_embed is None and e.old_action_unembeds is None
    for k in ('latents', 'rewards', 'values', 'lens', 'terminals'):
        opt['fine_' + k] = npy(getattr(e, k))
    opt['fine_log_probs'], opt['fine_actions'] = npy(e.log_probs.discrete), npy(e.actions.discrete)
    opt['fine_step_size'] = np.array(e.step_size)
    noise_dict('fine_', nz, opt)
    np.savez(os.path.join(OUT, 'options.npz'), **opt, **{'meta_' + k: np.array(v) for k, v in META.items()})

    np.savez(os.path.join(OUT, 'generate.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    print('generate margins', out['cached_margin'], out['nocache_margin'], out['prompt_margin'],
          'lens', out['cached_lens'], out['nocache_lens'], out['prompt_lens'])

    # ------------------------------------------------------------------ forward: parallel vs cached sequential
    out = {}
    g = torch.Generator().manual_seed(7)
    Tf = 4
    lat = torch.randn(B, Tf, 6, 8, generator=g)
    sig = torch.randint(0, 64, (B, Tf), generator=g)
    acts = torch.randint(0, 4, (B, Tf, 1), generator=g)
    # block-level intermediates of the same parallel forward, taken with forward hooks on the reference's own modules:
    # every layer hidden (the list the attention pools read, D4:3040/3172/3216), each pool's output, the final pool's
    # input / output and the learned-query pool that turns latents into spatial tokens -> blocks.npz
    blocks, hooks = {}, []
    tr = m.transformer
    def final_pool_hook(mod, args, kwargs, output):
        hid = kwargs['hiddens'] if 'hiddens' in kwargs else args[1]
        blocks['hiddens'] = npy(torch.stack(list(hid)))
        blocks['final_pool_in'] = npy(args[0]); blocks['final_pool_out'] = npy(output)
    hooks.append(tr.final_attn_pool.register_forward_hook(final_pool_hook, with_kwargs=True))
    for i, pool in enumerate(tr.attn_pools):
        if pool is None:
            continue
        hooks.append(pool.register_forward_hook(lambda mod, args, output, i=i: blocks.__setitem__(f'pool_out_{i}', npy(output))))
    hooks.append(m.latents_to_spatial_tokens.register_forward_hook(lambda mod, args, output: blocks.__setitem__('spatial_tokens', npy(output))))
    with torch.no_grad():
        pred, (emb, inter) = m(latents=lat, signal_levels=sig, step_sizes=4, discrete_actions=acts, latent_is_noised=True,
                               return_pred_only=True, return_intermediates=True)
        for hk in hooks:
            hk.remove()
        np.savez(os.path.join(OUT, 'blocks.npz'), **blocks, **{'meta_' + k: np.array(v) for k, v in META.items()})
        print('blocks:', {k: v.shape for k, v in blocks.items()})
        out.update(latents=npy(lat), signal_levels=npy(sig), actions=npy(acts), pred=npy(pred.flow[:, :, 0]),
                   agent_embed=npy(emb.agent[:, :, 0]), kv=npy(inter.main.next_kv_cache))
        tc, seq_agent, seq_pred = None, [], []
        for t in range(Tf):
            a = None if t == 0 else acts[:, t - 1:t]
            pred, (emb, tc) = m(latents=lat[:, t:t + 1], signal_levels=sig[:, t:t + 1], step_sizes=4, discrete_actions=a,
                                time_cache=tc, latent_is_noised=True, return_pred_only=True, return_intermediates=True)
            seq_agent.append(emb.agent[:, :, 0]); seq_pred.append(pred.flow[:, :, 0])
        out['seq_agent_embed'] = npy(torch.cat(seq_agent, 1)); out['seq_pred'] = npy(torch.cat(seq_pred, 1))
    print('parallel vs sequential (reference self-consistency):', np.abs(out['agent_embed'] - out['seq_agent_embed']).max())
    np.savez(os.path.join(OUT, 'forward.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})

    # ------------------------------------------------------------------ learn
    out = {}
    e = cached_exp
    heads = ('policy_head', 'value_head', 'action_embedder.discrete_action_unembed')
    for obj in ('ppo', 'spo', 'pmpo'):
        m.zero_grad()
        pl_, vl_ = m.learn_from_experience(e, objective=obj)
        pl_.backward(); vl_.backward()
        out[f'{obj}_policy_loss'], out[f'{obj}_value_loss'] = npy(pl_), npy(vl_)
        for k, p in m.named_parameters():
            if k.startswith(heads) and p.numel() > 0 and p.grad is not None:
                if obj == 'ppo' or p.ndim == 1 or 'unembed' in k:
                    out[f'{obj}_grad/{k}'] = npy(p.grad)
                else:       # large matrices of the other objectives: Frobenius norm + a strided sample
                    out[f'{obj}_gnorm/{k}'] = npy(p.grad.norm())
                    out[f'{obj}_gsample/{k}'] = npy(p.grad.flatten()[::97])
    # non-default learner options (kept in options.npz next to the non-default generate options)
    lopt = {}
    for tag, kw in (('nogate', dict(objective='ppo', use_delight_gating=False)),
                    ('temp', dict(objective='spo', delight_temperature=2.5)),
                    ('rawadv', dict(objective='ppo', normalize_advantages=False)),
                    ('pmpo_norm', dict(objective='pmpo', normalize_advantages=True, eps=1e-3))):
        m.zero_grad()
        pl_, vl_ = m.learn_from_experience(e, **kw)
        pl_.backward(); vl_.backward()
        lopt[f'learn_{tag}_policy_loss'], lopt[f'learn_{tag}_value_loss'] = npy(pl_), npy(vl_)
        for k, p in m.named_parameters():
            if k.startswith(heads) and p.numel() > 0 and p.grad is not None and (p.ndim == 1 or 'unembed' in k):
                lopt[f'learn_{tag}_grad/{k}'] = npy(p.grad)
    o_ = dict(np.load(os.path.join(OUT, 'options.npz')))
    o_.update(lopt)
    np.savez(os.path.join(OUT, 'options.npz'), **o_)
    # GAE with terminations / truncations
    g = torch.Generator().manual_seed(9)
    r = torch.randn(4, 7, generator=g); v = torch.randn(4, 7, generator=g)
    lens = torch.tensor([7, 3, 5, 1]); trunc = torch.tensor([True, False, False, True]); term = ~trunc
    ar = torch.arange(7)
    gae_masks = ar < (lens - 1).clamp(min=0)[:, None]
    term_seq = D4.flags_to_sequence(term, (lens - 1).clamp(min=0), 7)
    gae_masks = gae_masks.masked_fill(term_seq, False)
    len_mask = ar < lens[:, None]
    learn_mask = ar < (lens - trunc.long())[:, None]
    ret = D4.calc_gae(r.masked_fill(~len_mask, 0.), v.masked_fill(~len_mask, 0.), masks=gae_masks, learn_masks=learn_mask,
                      gamma=0.997, lam=0.95, use_accelerated=False)
    out.update(gae_rewards=npy(r), gae_values=npy(v), gae_lens=npy(lens), gae_trunc=npy(trunc), gae_term=npy(term), gae_returns=npy(ret))
    np.savez(os.path.join(OUT, 'learn.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})

    # ------------------------------------------------------------------ three DreamTrainer steps (trainers.py:1422-1452)
    out = {}
    m2 = build_reference_model(cfg, seed=0)
    with torch.no_grad():
        m2.action_embedder.discrete_action_unembed.mul_(0.3)
    popt = torch.optim.AdamW(m2.policy_head_parameters(), lr=3e-4, weight_decay=0.)
    vopt = torch.optim.AdamW(m2.value_head_parameters(), lr=3e-4, weight_decay=0.)
    Bt, Ht = 4, 4
    for step in range(3):
        nz = make_noise(cfg, Ht + 1, Bt, 200 + step)
        noise_dict(f'step{step}_', nz, out)
        with injected(nz):
            dreams = m2.generate(Ht + 1, batch_size=Bt, return_rewards_per_frame=True, return_agent_actions=True,
                                 return_log_probs_and_values=True)
        pl_, vl_ = m2.learn_from_experience(dreams, objective='ppo')
        pl_.backward()
        out[f'step{step}_policy_gnorm'] = npy(torch.nn.utils.clip_grad_norm_(m2.policy_head_parameters(), 0.5))
        popt.step(); popt.zero_grad()
        vl_.backward()
        out[f'step{step}_value_gnorm'] = npy(torch.nn.utils.clip_grad_norm_(m2.value_head_parameters(), 0.5))
        vopt.step(); vopt.zero_grad()
        out[f'step{step}_policy_loss'], out[f'step{step}_value_loss'] = npy(pl_), npy(vl_)
        out[f'step{step}_actions'] = npy(dreams.actions.discrete)
    for k, p in m2.named_parameters():
        if k.startswith(heads) and p.numel() > 0:
            if p.ndim == 1 or 'unembed' in k:
                out['final/' + k] = npy(p)
            else:
                out['final_sample/' + k] = npy(p.flatten()[::97])
                out['final_delta_norm/' + k] = npy((p - W[k]).norm())
    np.savez(os.path.join(OUT, 'trainer.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    # ------------------------------------------------------------------ a second architecture (variant.npz, weights_variant.npz)
    # two discrete action types, no register tokens, no tasks, one attention head, a time layer in every block,
    # two spatial tokens, single-token prediction: the options the main fixture leaves at one value
    cfg2 = Config(**CFG_VARIANT)
    mv = build_reference_model(cfg2, seed=3)
    with torch.no_grad():
        mv.action_embedder.discrete_action_unembed.mul_(0.3)
    Wv = weights_of(mv)
    np.savez(os.path.join(OUT, 'weights_variant.npz'), **{k: npy(v) for k, v in Wv.items() if v.numel() > 0},
             **{'meta_' + k: np.array(v) for k, v in META.items()}, **{'cfg_' + k: np.array(v) for k, v in CFG_VARIANT.items()})
    out = {}
    Bv, Tv = 4, 5
    nz = make_noise(cfg2, Tv, Bv, 301)
    with injected(nz):
        e = mv.generate(Tv, batch_size=Bv, return_for_policy_optimization=True, num_steps=8)
    exp_dict('cached_', e, out); noise_dict('cached_', nz, out)
    out['cached_margin'] = np.array(min_margin_multi(e, nz, cfg2))
    nz = make_noise(cfg2, Tv, Bv, 302)
    with injected(nz):
        e2 = mv.generate(Tv, batch_size=Bv, return_for_policy_optimization=True, use_time_cache=False)
    exp_dict('nocache_', e2, out); noise_dict('nocache_', nz, out)
    out['nocache_margin'] = np.array(min_margin_multi(e2, nz, cfg2))
    for obj in ('ppo', 'pmpo'):
        mv.zero_grad()
        pl_, vl_ = mv.learn_from_experience(e, objective=obj)
        pl_.backward(); vl_.backward()
        out[f'{obj}_policy_loss'], out[f'{obj}_value_loss'] = npy(pl_), npy(vl_)
        for k, p in mv.named_parameters():
            if k.startswith(heads) and p.numel() > 0 and p.grad is not None and (p.ndim == 1 or 'unembed' in k):
                out[f'{obj}_grad/{k}'] = npy(p.grad)
    np.savez(os.path.join(OUT, 'variant.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    print('variant margins', out['cached_margin'], out['nocache_margin'], 'lens', out['cached_lens'], out['nocache_lens'])

    # ------------------------------------------------------------------ one spatial token per latent token (samelen.npz)
    # num_spatial_tokens == num_latent_tokens switches latents_to_spatial_tokens to a Linear and drops the learned-query pool
    # of to_latent_pred (D4:4816-4834); this is BASELINE config 4's shape (4 latent tokens x 16)
    cfg3 = Config(**CFG_SAMELEN)
    ms = build_reference_model(cfg3, seed=5)
    with torch.no_grad():
        ms.action_embedder.discrete_action_unembed.mul_(0.3)
    Ws = weights_of(ms)
    np.savez(os.path.join(OUT, 'weights_samelen.npz'), **{k: npy(v) for k, v in Ws.items() if v.numel() > 0},
             **{'meta_' + k: np.array(v) for k, v in META.items()}, **{'cfg_' + k: np.array(v) for k, v in CFG_SAMELEN.items()})
    out = {}
    Bs, Ts = 3, 4
    nz = make_noise(cfg3, Ts, Bs, 401)
    with injected(nz):
        e = ms.generate(Ts, batch_size=Bs, return_for_policy_optimization=True)
    exp_dict('cached_', e, out); noise_dict('cached_', nz, out)
    out['cached_margin'] = np.array(min_margin(e, nz, cfg3))
    # env-wrapper pattern (dreamer4/env.py:445-483): one new frame per call, prompt = history, time cache carried
    nz = make_noise(cfg3, 3, Bs, 402)
    noise_dict('env_', nz, out)
    lat_hist = torch.zeros(Bs, 0, 4, 16); act_hist = torch.zeros(Bs, 0, 1, dtype=torch.long); tc = None
    for i in range(3):
        sub = {k: v[i:i + 1] for k, v in nz.items()}
        kw = dict(prompt_latents=lat_hist, prompt_discrete_actions=act_hist) if i > 0 else {}
        with injected(sub):
            e, tc = ms.generate(i + 1, batch_size=Bs, return_rewards_per_frame=True, return_agent_actions=True,
                                return_log_probs_and_values=True, time_cache=tc, return_time_cache=True, return_terminals=False, **kw)
        lat_hist, act_hist = e.latents, e.actions.discrete
        out[f'env{i}_latents'], out[f'env{i}_actions'], out[f'env{i}_values'] = npy(e.latents), npy(e.actions.discrete), npy(e.values)
    np.savez(os.path.join(OUT, 'samelen.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    print('samelen margin', out['cached_margin'], 'lens', out['cached_lens'])

    # ------------------------------------------------------------------ head dim 16 (headdim16.npz), the reference tests' own setting
    cfg4 = Config(**CFG_HEADDIM16)
    mh = build_reference_model(cfg4, seed=7)
    with torch.no_grad():
        mh.action_embedder.discrete_action_unembed.mul_(0.3)
    Wh = weights_of(mh)
    np.savez(os.path.join(OUT, 'weights_headdim16.npz'), **{k: npy(v) for k, v in Wh.items() if v.numel() > 0},
             **{'meta_' + k: np.array(v) for k, v in META.items()}, **{'cfg_' + k: np.array(v) for k, v in CFG_HEADDIM16.items()})
    out = {}
    nz = make_noise(cfg4, 5, 3, 501)
    with injected(nz):
        e = mh.generate(5, batch_size=3, return_for_policy_optimization=True)
    exp_dict('cached_', e, out); noise_dict('cached_', nz, out)
    out['cached_margin'] = np.array(min_margin(e, nz, cfg4))
    nz = make_noise(cfg4, 4, 3, 502)
    with injected(nz):
        e2, tc = mh.generate(4, batch_size=3, return_for_policy_optimization=True, use_time_cache=True, return_time_cache=True, return_terminals=False)
    exp_dict('tc_', e2, out); noise_dict('tc_', nz, out)
    out['tc_kv'] = npy(tc.main.next_kv_cache)
    np.savez(os.path.join(OUT, 'headdim16.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    print('headdim16 margin', out['cached_margin'], 'lens', out['cached_lens'], 'kv', out['tc_kv'].shape)

    # ------------------------------------------------------------------ action-free world model (actionfree.npz)
    cfg5 = Config(**CFG_ACTIONFREE)
    ma = build_reference_model(cfg5, seed=9)
    Wa = weights_of(ma)
    np.savez(os.path.join(OUT, 'weights_actionfree.npz'), **{k: npy(v) for k, v in Wa.items() if v.numel() > 0},
             **{'meta_' + k: np.array(v) for k, v in META.items()}, **{'cfg_' + k: np.array(v) for k, v in CFG_ACTIONFREE.items()})
    out = {}
    nz = make_noise(cfg5, 4, 3, 601)
    with injected(nz):
        lat = ma.generate(4, batch_size=3)
    out['plain_latents'] = npy(lat); noise_dict('plain_', nz, out)
    nz = make_noise(cfg5, 4, 3, 602)
    with injected(nz):
        e = ma.generate(4, batch_size=3, return_rewards_per_frame=True, return_terminals=True)
    out['rew_latents'], out['rew_rewards'], out['rew_agent_embed'] = npy(e.latents), npy(e.rewards), npy(e.agent_embed)
    out['rew_lens'], out['rew_terminals'] = npy(e.lens), npy(e.terminals)
    noise_dict('rew_', nz, out)
    np.savez(os.path.join(OUT, 'actionfree.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    print('actionfree lens', out['rew_lens'])

    # ------------------------------------------------------------------ non-default hyper-parameters (hyper.npz)
    cfg6 = Config(**CFG_HYPER)
    mp = build_reference_model(cfg6, seed=11)
    with torch.no_grad():
        mp.action_embedder.discrete_action_unembed.mul_(0.3)
    Wp = weights_of(mp)
    np.savez(os.path.join(OUT, 'weights_hyper.npz'), **{k: npy(v) for k, v in Wp.items() if v.numel() > 0},
             **{'meta_' + k: np.array(v) for k, v in META.items()}, **{'cfg_' + k: np.array(v) for k, v in CFG_HYPER.items()})
    out = {}
    nz = make_noise(cfg6, 6, 4, 701)
    with injected(nz):
        e = mp.generate(6, batch_size=4, return_for_policy_optimization=True, num_steps=2)
    exp_dict('cached_', e, out); noise_dict('cached_', nz, out)
    out['cached_margin'] = np.array(min_margin(e, nz, cfg6))
    for obj in ('ppo', 'spo', 'pmpo'):
        mp.zero_grad()
        pl_, vl_ = mp.learn_from_experience(e, objective=obj)
        pl_.backward(); vl_.backward()
        out[f'{obj}_policy_loss'], out[f'{obj}_value_loss'] = npy(pl_), npy(vl_)
        for k, p in mp.named_parameters():
            if k.startswith(heads) and p.numel() > 0 and p.grad is not None and (p.ndim == 1 or 'unembed' in k):
                out[f'{obj}_grad/{k}'] = npy(p.grad)
    np.savez(os.path.join(OUT, 'hyper.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    print('hyper margin', out['cached_margin'], 'lens', out['cached_lens'])

    # ------------------------------------------------------------------ a model without the terminal head (noterm.npz)
    cfg7 = Config(**CFG_NOTERM)
    mt = build_reference_model(cfg7, seed=13)
    with torch.no_grad():
        mt.action_embedder.discrete_action_unembed.mul_(0.3)
    Wt = weights_of(mt)
    assert not any(k.startswith('to_state_terminal_pred') for k in Wt)
    np.savez(os.path.join(OUT, 'weights_noterm.npz'), **{k: npy(v) for k, v in Wt.items() if v.numel() > 0},
             **{'meta_' + k: np.array(v) for k, v in META.items()}, **{'cfg_' + k: np.array(v) for k, v in CFG_NOTERM.items()})
    out = {}
    nz = make_noise(cfg7, 4, 3, 801)
    with injected(nz):
        e = mt.generate(4, batch_size=3, return_for_policy_optimization=True, tasks=torch.tensor([1, 0, 1]))
    exp_dict('cached_', e, out); noise_dict('cached_', nz, out)
    out['cached_margin'] = np.array(min_margin(e, nz, cfg7))
    mt.zero_grad()
    pl_, vl_ = mt.learn_from_experience(e, objective='ppo')
    out['ppo_policy_loss'], out['ppo_value_loss'] = npy(pl_), npy(vl_)
    np.savez(os.path.join(OUT, 'noterm.npz'), **out, **{'meta_' + k: np.array(v) for k, v in META.items()})
    print('noterm margin', out['cached_margin'], 'lens', out['cached_lens'], 'terminals', out['cached_terminals'])

    for fn in EXTRA.values():
        fn()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
