"""Drive the shim-imported reference with INJECTED noise (container only; test infrastructure).

The reference draws four random tensors per generated frame from torch's global
generator (D4:6475 randn, D4:6611 torch.bernoulli, MultiCategorical.sample's
uniform, D4:6670 randn_like).  `NoiseTape` replaces those four draws with
pre-drawn tensors so that the reference, `oracle/restate.py` and the HIP path can
be compared on identical inputs."""
from contextlib import contextmanager

import torch

from oracle.ref_import import load_reference
from oracle.restate import Config


def make_noise(cfg: Config, frames, batch, seed):
    """Same recipe as tests/util.make_noise (kept identical so fixtures can be regenerated from seeds)."""
    g = torch.Generator().manual_seed(seed)
    n, dl, A = cfg.num_latent_tokens, cfg.dim_latent, cfg.total_discrete_actions
    nz = dict(
        latent=torch.randn(frames, batch, n, dl, generator=g),
        context=torch.randn(frames, batch, n, dl, generator=g),
        gumbel_u=torch.rand(frames, batch, A, generator=g).clamp(1e-6, 1. - 1e-6),
        bern_u=torch.rand(frames, batch, generator=g),
    )
    nc = getattr(cfg, 'num_continuous_actions', 0)
    if nc > 0:      # Beta sampling as a ratio of Marsaglia-Tsang gammas: (normal, uniform) per rejection round
        nrm = torch.randn(frames, batch, nc, 2, 6, generator=g)
        uni = torch.rand(frames, batch, nc, 2, 6, generator=g).clamp(1e-6, 1. - 1e-6)
        nz['beta'] = torch.stack((nrm, uni), dim=-1)
    return nz


class NoiseTape:
    def __init__(self, noise):
        self.noise = noise
        self.f = 0
        self.a = 0      # action-type offset within a frame

    def randn(self, shape, **kw):
        t = self.noise['latent'][self.f]
        return t.reshape(shape).clone()

    def randn_like(self, t):
        out = self.noise['context'][self.f].reshape(t.shape).clone()
        self.f += 1
        self.a = 0
        return out

    def bernoulli(self, p):
        u = self.noise['bern_u'][self.f].reshape(p.shape)
        return (u < p).float()

    def uniform_like(self, t):
        n = t.shape[-1]
        u = self.noise['gumbel_u'][self.f][:, self.a:self.a + n].reshape(t.shape).clone()
        self.a += n
        return u

    def beta_noise_like(self, shape, device=None):
        t = self.noise['beta'][self.f]
        return t.reshape(*shape, *t.shape[-3:]).clone()


@contextmanager
def injected(noise):
    D4 = load_reference()
    import discrete_continuous_embed_readout as dcer
    from discrete_continuous_embed_readout import discrete_continuous_embed_readout as dcer_impl
    tape = NoiseTape(noise)
    saved = (D4.randn, D4.randn_like, torch.bernoulli, dcer.uniform_like, dcer_impl.beta_noise_like)
    D4.randn, D4.randn_like, torch.bernoulli, dcer.uniform_like = tape.randn, tape.randn_like, tape.bernoulli, tape.uniform_like
    dcer_impl.beta_noise_like = tape.beta_noise_like
    try:
        yield tape
    finally:
        D4.randn, D4.randn_like, torch.bernoulli, dcer.uniform_like, dcer_impl.beta_noise_like = saved


def build_reference_model(cfg: Config, seed=0, head_scale=True):
    """Reference DynamicsWorldModel for `cfg` with default init under `seed`; heads get
    non-trivial weights so logits/values are not all ~0."""
    D4 = load_reference()
    from x_mlps_pytorch import normed_mlp
    normed_mlp.RECIPE = cfg.head_mlp_recipe             # layer recipe of the stand-in normed MLP (restored below)
    from discrete_continuous_embed_readout import discrete_continuous_embed_readout as dcer_impl
    if cfg.continuous_beta_param != 'softplus_p1':
        assert hasattr(dcer_impl, 'BETA_PARAM'), 'continuous_beta_param is a switch of the stand-in package (oracle/shim), not of the real one'
    if hasattr(dcer_impl, 'BETA_PARAM'):
        dcer_impl.BETA_PARAM = cfg.continuous_beta_param    # link of the stand-in Beta head, read when the model builds its BetaDist
    torch.manual_seed(seed)
    m = D4.DynamicsWorldModel(
        num_continuous_actions=cfg.num_continuous_actions, reward_encoder_type=cfg.reward_encoder_type,
        dim=cfg.dim, dim_latent=cfg.dim_latent, num_latent_tokens=cfg.num_latent_tokens,
        depth=cfg.depth, time_block_every=cfg.time_block_every, attn_heads=cfg.attn_heads,
        attn_dim_head=cfg.attn_dim_head, num_spatial_tokens=cfg.num_spatial_tokens,
        num_register_tokens=cfg.num_register_tokens, max_steps=cfg.max_steps, num_tasks=cfg.num_tasks,
        num_discrete_actions=cfg.num_discrete_actions if len(cfg.num_discrete_actions) > 1 else (cfg.num_discrete_actions[0] if cfg.num_discrete_actions else 0),
        multi_token_pred_len=cfg.multi_token_pred_len,
        policy_head_mlp_depth=cfg.policy_head_mlp_depth, value_head_mlp_depth=cfg.value_head_mlp_depth,
        reward_encoder_kwargs=dict(num_bins=cfg.reward_num_bins, reward_range=cfg.reward_range),
        value_encoder_kwargs=dict(num_bins=cfg.value_num_bins, reward_range=cfg.value_range),
        predict_terminals=cfg.predict_terminals, attn_softclamp_value=cfg.attn_softclamp_value, gae_discount_factor=cfg.gae_discount_factor, gae_lambda=cfg.gae_lambda,
        ppo_eps_clip=cfg.ppo_eps_clip, policy_entropy_weight=cfg.policy_entropy_weight, use_delight_gating=cfg.use_delight_gating,
        delight_temperature=cfg.delight_temperature, pmpo_pos_to_neg_weight=cfg.pmpo_pos_to_neg_weight,
        pmpo_reverse_kl=cfg.pmpo_reverse_kl, pmpo_kl_div_loss_weight=cfg.pmpo_kl_div_loss_weight,
    ).eval()
    normed_mlp.RECIPE = 'pre_rms'
    if hasattr(dcer_impl, 'BETA_PARAM'):
        dcer_impl.BETA_PARAM = 'softplus_p1'
    post_ln = cfg.head_mlp_recipe == 'post_layer'
    if head_scale:
        with torch.no_grad():
            m.action_embedder.discrete_action_unembed.mul_(100.)
            m.action_embedder.continuous_action_unembed.mul_(30.)
            for p in m.to_reward_pred.parameters():
                if p.ndim == 3: p.mul_(20.)
            g = torch.Generator().manual_seed(seed + 1)
            for name, p in m.named_parameters():
                # learned tokens are randn*1e-2 by default: make them O(0.5) so they matter
                if name in ('register_tokens', 'agent_learned_embed', 'action_learned_embed') or name.endswith('queries'):
                    p.copy_(torch.randn(p.shape, generator=g) * 0.5)
                if name.endswith('gamma'):
                    p.copy_(torch.randn(p.shape, generator=g) * 0.2)
                if name.endswith('norm.weight') or name.endswith('norm_context.weight') or (name.endswith('.0.weight') and p.ndim == 1) \
                        or (post_ln and name.endswith('.1.weight') and p.ndim == 1):
                    p.copy_(1. + torch.randn(p.shape, generator=g) * 0.1)
            if cfg.predict_terminals:
                last = m.to_state_terminal_pred[0].layers[-1]
                (last if post_ln else last[1]).bias.fill_(-2.5)
    return m


def weights_of(model):
    return {k: v.detach().clone().float() for k, v in model.state_dict().items()}
