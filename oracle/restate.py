"""CPU fp32 restatement of dreamer4's imagination path (ORACLE — TEST INFRASTRUCTURE).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module, and only as the checker / the reported CPU baseline.  The
product path (`dreamer4_amd`) never routes through it.

What it restates (reference = /root/reference/dreamer4/dreamer4.py, "D4"):
  * DynamicsWorldModel.generate                 D4:6308-6774
  * DynamicsWorldModel.forward, inference branch D4:6792-7295
  * AxialSpaceTimeTransformer.forward           D4:2927-3267
  * Attention / naive_attend / rotary / K-norm  D4:1604-1756, 1968-2075
  * FeedForward, AttentionPool, LQAP            D4:2079-2210
  * ActionEmbedder embed/unembed/sample/logp    D4:1123-1562
  * HLGaussRewardEncoder                        D4:1041-1105
  * calc_gae, z_score, learn_from_experience    D4:404-410, 1566-1600, 5893-6305

Everything is a plain function of (config, weights dict keyed by the
reference's state_dict names, explicit noise tensors).  No hidden RNG.

Parity pin: `oracle/gen_golden.py` imports the reference unmodified through
`oracle/shim/` in the build container and freezes its outputs into tests/golden/*.npz;
tests/test_oracle_golden.py checks this file against those fixtures.  The
third-party pieces the reference delegates to (x_mlps_pytorch create_mlp /
Ensemble, hl_gauss_pytorch, discrete_continuous_embed_readout.MultiCategorical,
assoc_scan, torch_einops_utils.masked_mean) are absent from the image: they are
restated from their published algorithms and are PARITY UNPINNED against the
real packages (see DESIGN.md section "Oracle").
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

EPS_RMS = torch.finfo(torch.float32).eps  # nn.RMSNorm(eps=None) -> finfo(dtype).eps


# ----------------------------------------------------------------------------- config

@dataclass
class Config:
    """Supported-subset constructor arguments (names follow D4:4662-4778)."""
    dim: int
    dim_latent: int
    num_latent_tokens: int
    depth: int = 4
    time_block_every: int = 4
    attn_heads: int = 8
    attn_dim_head: int = 64
    attn_softclamp_value: float = 50.
    num_spatial_tokens: int = 4
    num_register_tokens: int = 8
    max_steps: int = 64
    num_tasks: int = 0
    num_discrete_actions: tuple = (4,)
    multi_token_pred_len: int = 8
    policy_head_mlp_depth: int = 3
    value_head_mlp_depth: int = 3
    terminal_mlp_depth: int = 1
    predict_terminals: bool = True
    reward_range: tuple = (-20., 20.)
    reward_num_bins: int = 255
    value_range: tuple = (-20., 20.)
    value_num_bins: int = 255
    hl_gauss_sigma_to_bin_ratio: float = 2.
    hl_gauss_eps: float = 1e-10
    pool_heads: int = 4          # AttentionPool defaults D4:2147-2148
    pool_dim_head: int = 64
    gae_discount_factor: float = 0.997
    gae_lambda: float = 0.95
    ppo_eps_clip: float = 0.2
    policy_entropy_weight: float = 0.01
    use_delight_gating: bool = True
    delight_temperature: float = 1.
    pmpo_pos_to_neg_weight: float = 0.5
    pmpo_reverse_kl: bool = True
    pmpo_kl_div_loss_weight: float = 0.3
    rotary_theta: float = 10000.
    num_continuous_actions: int = 0          # Beta policy head (continuous_dist_type='beta', D4:1131, 1172-1173)
    reward_encoder_type: str = 'hl_gauss'    # or 'symexp_two_hot' (D4:947-1040)
    head_mlp_recipe: str = 'pre_rms'         # layer recipe of x_mlps_pytorch's normed MLP: 'pre_rms' | 'post_layer' (see mlp())
    continuous_beta_param: str = 'softplus_p1'   # link of the Beta head's raw parameters: 'softplus_p1' | 'exp_p1' (see beta_alpha_beta())

    def __post_init__(self):
        if isinstance(self.num_discrete_actions, int):
            self.num_discrete_actions = (self.num_discrete_actions,)
        self.num_discrete_actions = tuple(int(n) for n in self.num_discrete_actions if n > 0)

    @property
    def is_time(self):
        return [((i + 1) % self.time_block_every) == 0 for i in range(self.depth)]

    @property
    def num_time_layers(self):
        return sum(self.is_time)

    @property
    def has_actions(self):
        return len(self.num_discrete_actions) > 0 or self.num_continuous_actions > 0

    @property
    def tokens_per_frame(self):
        # [flow | space | registers | action (only with an action space) | agent]   D4:7222, 7124-7130
        return 1 + self.num_spatial_tokens + self.num_register_tokens + (1 if self.has_actions else 0) + 1

    @property
    def ff_inner(self):
        return int(self.dim * 4 * 2 / 3)    # D4:2094

    @property
    def total_discrete_actions(self):
        return sum(self.num_discrete_actions)


# ----------------------------------------------------------------------------- small ops

def rmsnorm(x, w):
    return x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + EPS_RMS) * w


def l2norm(t):
    return F.normalize(t, dim=-1, p=2)      # x / max(||x||, 1e-12)   D4:521


def softclamp(t, value):
    return (t / value).tanh() * value       # D4:527


def rotary_freqs(cfg: Config, seq_len, offset, inv_freq=None):
    """D4:1604-1624."""
    dh = cfg.attn_dim_head
    if inv_freq is None:
        inv_freq = 1.0 / (cfg.rotary_theta ** (torch.arange(0, dh, 2).float() / dh))
    t = torch.arange(seq_len).float() + offset
    freqs = t[:, None] * inv_freq[None, :]
    return torch.cat((freqs, freqs), dim=-1)            # (n, dh)


def apply_rotations(rot, t):
    """D4:1626-1659; rot (n, dh), t (b, h, n, dh)."""
    n = t.shape[-2]
    if rot.shape[-2] > n:
        rot = rot[-n:]
    x1, x2 = t.chunk(2, dim=-1)
    half = torch.cat((-x2, x1), dim=-1)
    return t * rot.cos() + half * rot.sin()


def attend(q, k, v, softclamp_value=None, mask=None, causal=False):
    """naive_attend D4:1683-1756 (the SDPA branch computes the same math)."""
    scale = q.shape[-1] ** -0.5
    sim = torch.einsum('bhid,bhjd->bhij', q, k) * scale
    if softclamp_value is not None:
        sim = softclamp(sim, softclamp_value)
    neg = -torch.finfo(sim.dtype).max
    if mask is not None:
        sim = sim.masked_fill(~mask, neg)
    if causal:
        i, j = sim.shape[-2:]
        cm = torch.ones((i, j), dtype=torch.bool).triu(j - i + 1)
        sim = sim.masked_fill(cm, neg)
    attn = sim.softmax(dim=-1)
    return torch.einsum('bhij,bhjd->bhid', attn, v)


def special_token_mask(seq_len, num_special):
    """D4:1769-1783 with special_attend_only_itself=False: ordinary queries may not see special keys."""
    q = torch.arange(seq_len)[:, None]
    k = torch.arange(seq_len)[None, :]
    start = seq_len - num_special
    return ~((q < start) & (k >= start))


def attention(W, pre, tokens, *, heads, dim_head, context=None, kv_cache=None, rot=None,
              causal=False, residual_values=None, softclamp_value=None, mask=None,
              belief=True, has_ctx_norm=False):
    """Attention.forward D4:1968-2075.  tokens (b, n, d); returns (out, (k, v))."""
    x = rmsnorm(tokens, W[pre + 'norm.weight'])
    q = x @ W[pre + 'to_q.weight'].t()
    if context is not None:
        ctx = rmsnorm(context, W[pre + 'norm_context.weight']) if has_ctx_norm else context
    else:
        ctx = x
    k = ctx @ W[pre + 'to_k.weight'].t()
    v = ctx @ W[pre + 'to_v.weight'].t()

    def split(t):
        b, n, _ = t.shape
        return t.reshape(b, n, -1, dim_head).transpose(1, 2)       # b h n d

    q, k, v = split(q), split(k), split(v)

    if residual_values is not None:                                    # (b, n, h, d)
        rv = residual_values.transpose(1, 2)
        mix = torch.sigmoid(x @ W[pre + 'to_learned_value_residual_mix.0.weight'].t()
                            + W[pre + 'to_learned_value_residual_mix.0.bias'])
        mix = mix.transpose(1, 2)[..., None]                           # b h n 1
        v = v.lerp(rv, mix)

    gamma = W[pre + 'k_heads_rmsnorm.gamma']                           # (h, d)
    k = l2norm(k) * ((gamma + 1.) * dim_head ** 0.5)[None, :, None, :]

    if rot is not None:
        q = apply_rotations(rot, q)
        k = apply_rotations(rot, k)

    v_for_belief = v if (belief and context is None) else None

    if kv_cache is not None:
        ck, cv = kv_cache
        k = torch.cat((ck, k), dim=-2)
        v = torch.cat((cv, v), dim=-2)

    out = attend(q, k, v, softclamp_value=softclamp_value, mask=mask, causal=causal)

    if v_for_belief is not None:
        vn = l2norm(v_for_belief)
        out = out - (out * vn).sum(dim=-1, keepdim=True) * vn

    gates = torch.sigmoid(x @ W[pre + 'to_gates.0.weight'].t())        # b n h
    out = out * gates.transpose(1, 2)[..., None]

    b, h, n, d = out.shape
    out = out.transpose(1, 2).reshape(b, n, h * d)
    out = out @ W[pre + 'to_out.weight'].t()
    return out, (k, v)


def feedforward(W, pre, x):
    """FeedForward.forward D4:2105-2116 (SiLU-GLU: first chunk value, second gate)."""
    h = rmsnorm(x, W[pre + 'norm.weight'])
    h = h @ W[pre + 'proj_in.weight'].t() + W[pre + 'proj_in.bias']
    a, g = h.chunk(2, dim=-1)
    h = a * F.silu(g)
    return h @ W[pre + 'proj_out.weight'].t() + W[pre + 'proj_out.bias']


def attention_pool(cfg, W, pre, x, hiddens):
    """Residual(AttentionPool) D4:2143-2177 + 1869: one query per token over the stack of layer hiddens."""
    shape = x.shape
    ctx = torch.stack(hiddens, dim=-2).reshape(-1, len(hiddens), shape[-1])
    q = x.reshape(-1, 1, shape[-1])
    out, _ = attention(W, pre + 'fn.attn.', q, heads=cfg.pool_heads, dim_head=cfg.pool_dim_head,
                       context=ctx, belief=False, has_ctx_norm=True)
    return x + out.reshape(shape)


def lq_attn_pool(cfg, W, pre, x):
    """LearnedQueriesAttentionPool D4:2179-2210.  x (..., n, d_kv) -> (..., num_queries, dim)."""
    lead = x.shape[:-2]
    ctx = x.reshape(-1, *x.shape[-2:])
    queries = W[pre + 'queries'][None].expand(ctx.shape[0], -1, -1)
    out, _ = attention(W, pre + 'attn.', queries, heads=cfg.attn_heads, dim_head=cfg.attn_dim_head,
                       context=ctx, belief=False, has_ctx_norm=True)
    return out.reshape(*lead, *out.shape[-2:])


# ----------------------------------------------------------------------------- trunk

@dataclass
class TrunkCache:
    kv: list = field(default_factory=list)     # per time layer: (k, v) each (b*s, h, t, dh)
    token_count: int = 0


def transformer(cfg: Config, W, tokens, cache: TrunkCache | None = None, pre='transformer.', trace: dict | None = None, num_special=1):
    """AxialSpaceTimeTransformer.forward D4:2927-3267 (defaults: value residual, attn pools,
    final special cross-attn; the final RMSNorm is applied when the weights hold one — the dynamics model builds its trunk
    with final_norm=False, the tokenizer's decoder with the default True).  tokens (b, t, s, d).  When a non-empty cache is
    given and t > 1 only the last frame is processed (D4:2960-2961)."""
    b, t, s, d = tokens.shape
    h, dh = cfg.attn_heads, cfg.attn_dim_head
    has_cache = cache is not None and len(cache.kv) > 0
    token_count = cache.token_count if cache is not None else 0
    if has_cache and t > 1:
        tokens = tokens[:, -1:]
        t = 1

    space_mask = special_token_mask(s, num_special)
    rot = rotary_freqs(cfg, t, token_count, W.get(pre + 'time_rotary.inv_freq'))

    vres = rmsnorm(tokens, W[pre + 'to_value_residual.0.weight']) @ W[pre + 'to_value_residual.1.weight'].t()
    vres = vres.reshape(b, t, s, h, dh)

    layer_hiddens = [tokens]
    new_kv = []
    time_idx = 0
    for i, is_time in enumerate(cfg.is_time):
        ap = f'{pre}layers.{i}.2.fn.'
        if is_time:
            x = tokens.transpose(1, 2).reshape(b * s, t, d)
            rv = vres.transpose(1, 2).reshape(b * s, t, h, dh)
            kvc = cache.kv[time_idx] if has_cache else None
            out, kv = attention(W, ap, x, heads=h, dim_head=dh, kv_cache=kvc, rot=rot, causal=True,
                                residual_values=rv, softclamp_value=cfg.attn_softclamp_value)
            new_kv.append(kv)
            time_idx += 1
            out = out.reshape(b, s, t, d).transpose(1, 2)
        else:
            x = tokens.reshape(b * t, s, d)
            rv = vres.reshape(b * t, s, h, dh)
            out, _ = attention(W, ap, x, heads=h, dim_head=dh, residual_values=rv,
                               softclamp_value=cfg.attn_softclamp_value, mask=space_mask)
            out = out.reshape(b, t, s, d)
        tokens = tokens + out
        layer_hiddens.append(tokens)

        tokens = tokens + feedforward(W, f'{pre}layers.{i}.3.fn.', tokens)
        layer_hiddens.append(tokens)

        if i != cfg.depth - 1:
            tokens = attention_pool(cfg, W, f'{pre}attn_pools.{i}.', tokens, layer_hiddens)
            if trace is not None:
                trace[f'pool_out_{i}'] = tokens

    # agent token cross-attends the non-special tokens of its frame   D4:3227-3238
    non_special, special = tokens[:, :, :-num_special], tokens[:, :, -num_special:]
    cp = pre + 'final_special_cross_attn.fn.'
    q = special.reshape(b * t, num_special, d)
    ctx = non_special.reshape(b * t, s - num_special, d)
    out, _ = attention(W, cp, q, heads=h, dim_head=dh, context=ctx, belief=True, has_ctx_norm=True)
    special = special + out.reshape(b, t, num_special, d)
    special = special + feedforward(W, pre + 'final_special_ff.fn.', special)
    tokens = torch.cat((non_special, special), dim=2)

    if trace is not None:
        trace['hiddens'] = torch.stack(layer_hiddens)
        trace['final_pool_in'] = tokens
    tokens = attention_pool(cfg, W, pre + 'final_attn_pool.', tokens, layer_hiddens)
    if trace is not None:
        trace['final_pool_out'] = tokens
    if pre + 'final_norm.weight' in W:                                   # D4:3246
        tokens = rmsnorm(tokens, W[pre + 'final_norm.weight'])

    new_cache = TrunkCache(kv=new_kv, token_count=token_count + t)
    return tokens, new_cache


# ----------------------------------------------------------------------------- world-model forward

def action_tokens(cfg: Config, W, actions, time, batch, cont_actions=None):
    """D4:7088-7126 + ActionEmbedder.forward D4:1501-1562 (all action types; discrete: embedding gather summed over types,
    continuous: embed[type] * value summed over types, D4:1535-1545).
    actions (b, time-1 | time, na) int64 or None, cont_actions (b, same, nc) float or None -> (b, time, d); frame 0 gets a zero token."""
    d = cfg.dim
    have_d = actions is not None and actions.shape[1] > 0 and actions.shape[-1] > 0
    have_c = cont_actions is not None and cont_actions.shape[1] > 0 and cont_actions.shape[-1] > 0
    if not (have_d or have_c):
        return torch.zeros(batch, time, d)
    emb = 0.
    if have_d:
        offsets = torch.tensor([0, *torch.tensor(cfg.num_discrete_actions).cumsum(0)[:-1].tolist()])
        emb = emb + W['action_embedder.discrete_action_embed.weight'][actions + offsets].sum(dim=-2)
    if have_c:
        emb = emb + (W['action_embedder.continuous_action_embed.weight'] * cont_actions[..., None]).sum(dim=-2)
    emb = emb + W['action_learned_embed']                              # (1, d) broadcast
    if emb.shape[1] == time:
        emb = emb[:, :-1]
    assert emb.shape[1] == time - 1
    return F.pad(emb, (0, 0, 1, 0), value=0.)


def wm_forward(cfg: Config, W, latents, signal_levels, step_size, actions=None, tasks=None,
               cache: TrunkCache | None = None, trace: dict | None = None, cont_actions=None):
    """DynamicsWorldModel.forward(latent_is_noised=True, return_pred_only=True,
    return_intermediates=True)  D4:6792-7295.

    latents (b, t, n, dl); signal_levels (b, t) int64; step_size python int (power of two), or an int64 tensor (b,) of
    step_sizes_log2 (the training branch samples one per trajectory).
    Returns pred (b, t', n, dl), agent_embed (b, t', d), new cache, where t' = 1 when a
    non-empty cache was supplied (only the last frame is evaluated — per-frame modules are
    independent across frames, D4:7168/7251, so the dropped frames are never consumed)."""
    b, t, n, dl = latents.shape
    d = cfg.dim
    has_cache = cache is not None and len(cache.kv) > 0
    act_tok = action_tokens(cfg, W, actions, t, b, cont_actions)
    if has_cache and t > 1:
        latents, signal_levels, act_tok = latents[:, -1:], signal_levels[:, -1:], act_tok[:, -1:]
        t = 1

    same_len = cfg.num_spatial_tokens == cfg.num_latent_tokens                      # D4:4816-4834: Linear in, no pool out
    if same_len:
        space = latents @ W['latents_to_spatial_tokens.weight'].t() + W['latents_to_spatial_tokens.bias']
    else:
        space = lq_attn_pool(cfg, W, 'latents_to_spatial_tokens.', latents)         # b t ns d
    sig = W['signal_levels_embed.weight'][signal_levels]                             # b t d/2
    if torch.is_tensor(step_size):                                                   # training: step_sizes_log2 (b,) per trajectory  D4:7183-7184
        stp = W['step_size_embed.weight'][step_size][:, None].expand(b, t, -1)
    else:
        stp = W['step_size_embed.weight'][int(math.log2(step_size))].expand(b, t, -1)
    flow_tok = torch.cat((sig, stp), dim=-1)[:, :, None]
    regs = (W['register_tokens'] if cfg.num_register_tokens > 0 else latents.new_zeros(0, d)).expand(b, t, -1, -1)
    agent = W['agent_learned_embed'].expand(b, -1, -1)                               # b 1 d
    if tasks is not None:
        agent = agent + W['task_embed.weight'][tasks][:, None]
    agent = agent[:, None].expand(b, t, -1, -1)

    if not cfg.has_actions:                                                         # no action space: no action token  D4:7128-7130
        tokens = torch.cat((flow_tok, space, regs, agent), dim=2)
    else:
        tokens = torch.cat((flow_tok, space, regs, act_tok[:, :, None], agent), dim=2)
    if trace is not None:
        trace['spatial_tokens'] = space
    tokens, new_cache = transformer(cfg, W, tokens, cache, trace=trace)

    ns = cfg.num_spatial_tokens
    space_out = tokens[:, :, 1:1 + ns]
    agent_embed = tokens[:, :, -1]

    x = rmsnorm(space_out, W['to_latent_pred.0.weight'])
    if not same_len:
        x = lq_attn_pool(cfg, W, 'to_latent_pred.1.', x)
    pred = x @ W['to_latent_pred.2.weight'].t()
    return pred, agent_embed, new_cache


# ----------------------------------------------------------------------------- dynamics training losses (flow + shortcut)

def ramp_weight(times, slope=0.9, intercept=0.1):
    return slope * times + intercept                                                 # D4:897-899, eq. (8)


def lens_to_mask(lens, time):
    return torch.arange(time)[None, :] < lens[:, None]                               # D4 `lens_to_mask`


def dynamics_flow_losses(cfg: Config, W, latents, noise, signal_levels, step_sizes_log2, shortcut_train, actions=None, tasks=None,
                         cont_actions=None, lens=None):
    """The flow and shortcut-consistency losses of DynamicsWorldModel.forward in training (D4:6990-7003, 7335-7431; x-space prediction,
    the default `pred_orig_latent=True`; no proprio, no variable lengths, no loss normalisers — the dynamics model's defaults).
    latents (b, t, n, dl) data, noise the same shape, signal_levels (b, t), step_sizes_log2 (b,) int64, shortcut_train: the coin of
    D4:6965.  Returns (flow_loss, shortcut_loss)."""
    times = signal_levels.float() / cfg.max_steps                                    # D4:5413
    tt = times[:, :, None, None]
    noised = noise.lerp(latents, tt)                                                 # D4:7003
    pred = wm_forward(cfg, W, noised, signal_levels, step_sizes_log2, actions=actions, tasks=tasks, cont_actions=cont_actions)[0]
    flow_losses = F.mse_loss(pred, latents, reduction='none') * ramp_weight(times)[:, :, None, None]      # D4:7350, 7410-7414
    sel = (lambda x: x[lens_to_mask(lens, latents.shape[1])]) if lens is not None else (lambda x: x)          # variable lengths  D4:7418-7426
    if not shortcut_train:
        return sel(flow_losses).mean(), latents.new_zeros(())
    with torch.no_grad():                                                            # D4:7313, 7356-7388
        half_log2 = step_sizes_log2 - 1
        half = 2 ** half_log2
        first = wm_forward(cfg, W, noised, signal_levels, half_log2, actions=actions, tasks=tasks, cont_actions=cont_actions)[0]
        first_flow = (first - noised) / (1. - tt)
        denoised = noised + first_flow * (half[:, None, None, None] / cfg.max_steps)
        sig2 = signal_levels + half[:, None]
        second = wm_forward(cfg, W, denoised, sig2, half_log2, actions=actions, tasks=tasks, cont_actions=cont_actions)[0]
        second_flow = (second - denoised) / (1. - (sig2.float() / cfg.max_steps)[:, :, None, None])
        target = (first_flow + second_flow) / 2
    shortcut_pred = (pred - noised) / (1. - tt)                                      # D4:7397-7398
    shortcut_losses = F.mse_loss(shortcut_pred, target, reduction='none') * (1. - tt) ** 2
    return sel(flow_losses).mean(), sel(shortcut_losses).mean()


# ----------------------------------------------------------------------------- heads

def layernorm(x, w, b, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = (x - mu).pow(2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def mlp(W, pre, x, n_layers, recipe='pre_rms'):
    """Normed MLP of x_mlps_pytorch.create_mlp (D4:4950, 5083, 5095).  The package is absent: the layer recipe is a descriptor
    (oracle/shim/x_mlps_pytorch/normed_mlp.py, PARITY UNPINNED):
      'pre_rms'     RMSNorm -> Linear -> SiLU, no activation after the last Linear
      'post_layer'  Linear -> LayerNorm -> SiLU, the last layer a bare Linear"""
    for i in range(n_layers):
        last = i == n_layers - 1
        if recipe == 'pre_rms':
            x = rmsnorm(x, W[f'{pre}layers.{i}.0.weight'])
            x = x @ W[f'{pre}layers.{i}.1.weight'].t() + W[f'{pre}layers.{i}.1.bias']
            if not last:
                x = F.silu(x)
        elif last:
            x = x @ W[f'{pre}layers.{i}.weight'].t() + W[f'{pre}layers.{i}.bias']
        else:
            x = x @ W[f'{pre}layers.{i}.0.weight'].t() + W[f'{pre}layers.{i}.0.bias']
            x = F.silu(layernorm(x, W[f'{pre}layers.{i}.1.weight'], W[f'{pre}layers.{i}.1.bias']))
    return x


def mlp_num_layers(depth):
    return depth + 2       # create_mlp widths (dim_in, dim x (depth+1), dim_out)


def hl_gauss_centers(vrange, num_bins):
    support = torch.linspace(vrange[0], vrange[1], num_bins + 1).float()
    return support, (support[:-1] + support[1:]) / 2


def hl_gauss_to_scalar(logits, vrange, num_bins):
    _, centers = hl_gauss_centers(vrange, num_bins)
    return (logits.softmax(dim=-1) * centers).sum(dim=-1)


def symexp_bin_values(vrange, num_bins):
    v = torch.linspace(vrange[0], vrange[1], num_bins)
    return v.sign() * (torch.exp(v.abs()) - 1.)                  # D4:958-960


def bins_to_scalar(cfg, logits, vrange, num_bins):
    """reward / value encoder .bins_to_scalar_value: HLGaussRewardEncoder D4:1088-1096 or SymExpTwoHot D4:987-993."""
    if cfg.reward_encoder_type == 'symexp_two_hot':
        return (logits.softmax(dim=-1) * symexp_bin_values(vrange, num_bins)).sum(dim=-1)
    return hl_gauss_to_scalar(logits, vrange, num_bins)


def symexp_two_hot(values, vrange, num_bins):
    """SymExpTwoHot.forward D4:995-1040."""
    bv = symexp_bin_values(vrange, num_bins)
    shape = values.shape
    v = values.reshape(-1).clamp(min=bv[0], max=bv[-1])
    idx = torch.searchsorted(bv, v)
    li = (idx - 1).clamp(min=0)
    ri = (li + 1).clamp(max=num_bins - 1)
    lv, rv = bv[li], bv[ri]
    wl = (rv - v) / (rv - lv)
    enc = torch.zeros(v.shape[0], num_bins)
    enc.scatter_(-1, li[:, None], wl[:, None])
    enc.scatter_(-1, ri[:, None], (1. - wl)[:, None])
    return enc.reshape(*shape, num_bins)


def hl_gauss_to_probs(cfg: Config, values, vrange, num_bins):
    support, _ = hl_gauss_centers(vrange, num_bins)
    sigma = cfg.hl_gauss_sigma_to_bin_ratio * (vrange[1] - vrange[0]) / num_bins
    values = values.clamp(vrange[0], vrange[1])
    cdf = torch.special.erf((support - values[..., None]) / (math.sqrt(2.) * sigma))
    z = cdf[..., -1] - cdf[..., 0]
    return (cdf[..., 1:] - cdf[..., :-1]) / z.clamp(min=cfg.hl_gauss_eps)[..., None]


def reward_head(cfg, W, agent_embed):
    """to_reward_pred.forward_one(x, id=0) -> bins_to_scalar  D4:5067-5075, 6598-6599."""
    x = rmsnorm(agent_embed, W['to_reward_pred.params.0'][0])
    logits = x @ W['to_reward_pred.params.1'][0].t()
    return bins_to_scalar(cfg, logits, cfg.reward_range, cfg.reward_num_bins)


def terminal_prob(cfg, W, denoised_latent):
    """D4:6606-6611: mean over (view, latent token) -> MLP -> sigmoid."""
    pooled = denoised_latent.mean(dim=-2)
    logit = mlp(W, 'to_state_terminal_pred.0.', pooled, mlp_num_layers(cfg.terminal_mlp_depth), cfg.head_mlp_recipe)
    return logit.squeeze(-1).sigmoid()


def policy_logits(cfg, W, policy_embed):
    """ActionEmbedder.unembed(pred_head_index=0) D4:1313-1326 -> (..., total_discrete_actions)."""
    un = W['action_embedder.discrete_action_unembed'][:, 0]            # (na_total, 4d)
    return policy_embed @ un.t()


def _log(t, eps=1e-20):
    return t.clamp(min=eps).log()


def sample_discrete(cfg, logits, u, temperature=1.):
    """Gumbel-max per action type (D4:485-497 / MultiCategorical.sample).  u uniform, same shape as logits."""
    outs, o = [], 0
    for n in cfg.num_discrete_actions:
        l, uu = logits[..., o:o + n], u[..., o:o + n]
        g = -_log(-_log(uu))
        outs.append((l / max(temperature, 1e-10) + g).argmax(dim=-1))
        o += n
    return torch.stack(outs, dim=-1)


def discrete_log_probs(cfg, logits, actions, with_entropy=False):
    lps, ents, o = [], [], 0
    for i, n in enumerate(cfg.num_discrete_actions):
        lp = logits[..., o:o + n].log_softmax(dim=-1)
        lps.append(lp.gather(-1, actions[..., i:i + 1]).squeeze(-1))
        ents.append(-(lp.exp() * lp).sum(dim=-1))
        o += n
    lps = torch.stack(lps, dim=-1)
    return (lps, torch.stack(ents, dim=-1)) if with_entropy else lps


# ---- continuous actions: Beta policy head (Readout / BetaDist of discrete_continuous_embed_readout, stood in by
# oracle/shim/discrete_continuous_embed_readout — parameterisation, tempering and sampler ASSUMED there, PARITY UNPINNED)

GAMMA_ROUNDS = 6


def policy_cont_params(cfg, W, policy_embed):
    """ActionEmbedder.unembed(pred_head_index=0), continuous half D4:1340-1353: (..., nc, 2) raw parameters."""
    un = W['action_embedder.continuous_action_unembed'][:, 0]         # (nc, 4d, 2)
    return torch.einsum('...d,ndt->...nt', policy_embed, un)


def beta_alpha_beta(params, kind='softplus_p1'):
    """alpha, beta of BetaDist(unimodal=True) D4:1172-1173: link(raw) + 1.  The link lives in the third-party package (absent, unpinned):
    a descriptor, like the MLP recipe — softplus (the stand-in's default) or exp."""
    assert kind in ('softplus_p1', 'exp_p1'), kind
    link = torch.exp if kind == 'exp_p1' else F.softplus
    return link(params[..., 0]) + 1., link(params[..., 1]) + 1.


def gamma_from_noise(shape_param, noise):
    """Marsaglia-Tsang with the rejection loop unrolled over the injected rounds; noise (..., rounds, 2) = (normal, uniform)."""
    d = shape_param - 1. / 3.
    c = 1. / torch.sqrt(9. * d)
    out, done = None, torch.zeros_like(shape_param, dtype=torch.bool)
    rounds = noise.shape[-2]
    for r in range(rounds):
        x, u = noise[..., r, 0], noise[..., r, 1]
        t = 1. + c * x
        v = t * t * t
        ok = (v > 0.) & (torch.log(u.clamp(min=1e-30)) < 0.5 * x * x + d - d * v + d * torch.log(v.clamp(min=1e-30)))
        cand = d * v
        take = (ok | (r == rounds - 1)) & ~done
        out = cand if out is None else torch.where(take, cand, out)
        done = done | ok
    return out.clamp(min=1e-30)


def beta_accept_margin(params, noise, temperature=1., kind='softplus_p1'):
    """Smallest |log u - bound| over the rejection rounds that decided a draw: exact agreement of the accept / reject decisions
    between fp32 implementations is only well posed when this is comfortably above their rounding differences."""
    a, b = beta_alpha_beta(params, kind)
    t = max(float(temperature), 1e-10)
    best = float('inf')
    for shape, nz in ((1. + (a - 1.) / t, noise[..., 0, :, :]), (1. + (b - 1.) / t, noise[..., 1, :, :])):
        d = shape - 1. / 3.
        c = 1. / torch.sqrt(9. * d)
        done = torch.zeros_like(shape, dtype=torch.bool)
        for r in range(nz.shape[-2]):
            x, u = nz[..., r, 0], nz[..., r, 1]
            v = (1. + c * x) ** 3
            gap = (torch.log(u.clamp(min=1e-30)) - (0.5 * x * x + d - d * v + d * torch.log(v.clamp(min=1e-30)))).abs()
            gap = torch.where(v > 0., gap, v.abs())
            live = ~done
            if live.any():
                best = min(best, float(gap[live].min()))
            done = done | ((v > 0.) & (torch.log(u.clamp(min=1e-30)) < 0.5 * x * x + d - d * v + d * torch.log(v.clamp(min=1e-30))))
    return best


def sample_continuous(params, noise, temperature=1., kind='softplus_p1'):
    """Readout.sample_continuous D4:1379-1383: Beta(1 + (alpha-1)/T, 1 + (beta-1)/T) as a ratio of gammas; noise (..., nc, 2, rounds, 2)."""
    a, b = beta_alpha_beta(params, kind)
    t = max(float(temperature), 1e-10)
    a, b = 1. + (a - 1.) / t, 1. + (b - 1.) / t
    ga, gb = gamma_from_noise(a, noise[..., 0, :, :]), gamma_from_noise(b, noise[..., 1, :, :])
    return ga / (ga + gb)


def beta_log_prob(params, x, kind='softplus_p1'):
    a, b = beta_alpha_beta(params, kind)
    return (a - 1.) * torch.log(x) + (b - 1.) * torch.log1p(-x) + torch.lgamma(a + b) - torch.lgamma(a) - torch.lgamma(b)


def beta_entropy(params, kind='softplus_p1'):
    a, b = beta_alpha_beta(params, kind)
    lbeta = torch.lgamma(a) + torch.lgamma(b) - torch.lgamma(a + b)
    return lbeta - (a - 1.) * torch.digamma(a) - (b - 1.) * torch.digamma(b) + (a + b - 2.) * torch.digamma(a + b)


def beta_kl(p_params, q_params, kind='softplus_p1'):
    """KL(Beta_p || Beta_q)  (torch.distributions.kl._kl_beta_beta)."""
    a1, b1 = beta_alpha_beta(p_params, kind)
    a2, b2 = beta_alpha_beta(q_params, kind)
    lb = lambda a, b: torch.lgamma(a) + torch.lgamma(b) - torch.lgamma(a + b)
    return (lb(a2, b2) - lb(a1, b1) + (a1 - a2) * torch.digamma(a1) + (b1 - b2) * torch.digamma(b1)
            + (a2 - a1 + b2 - b1) * torch.digamma(a1 + b1))


def policy_head(cfg, W, agent_embed):
    return mlp(W, 'policy_head.', agent_embed, mlp_num_layers(cfg.policy_head_mlp_depth), cfg.head_mlp_recipe)


def value_head_bins(cfg, W, agent_embed):
    return mlp(W, 'value_head.', agent_embed, mlp_num_layers(cfg.value_head_mlp_depth), cfg.head_mlp_recipe)


# ----------------------------------------------------------------------------- dynamics training losses (agent token)

def mtp_targets(t, steps):
    """create_multi_token_prediction_targets D4:530-552: t (b, n, ...) -> (b, n, steps, ...), mask (b, n, steps); out-of-range -> index 0."""
    n = t.shape[1]
    idx = torch.arange(n)[:, None] + torch.arange(steps)[None, :]
    mask = idx < n
    idx = idx.masked_fill(~mask, 0)
    return t[:, idx], mask[None].expand(t.shape[0], -1, -1)


def dynamics_agent_losses(cfg: Config, W, agent_embed, latents, rewards=None, actions=None, terminals=None, lens=None, cont_actions=None):
    """The agent-token losses of the training forward (D4:7432-7598): multi-token-prediction reward cross entropy against the
    encoder's soft targets, terminal BCE with DreamerV3 label smoothing, behaviour-cloning log-likelihood of the discrete actions
    (multi-token prediction, `shift_action_tokens=True`).  agent_embed (b, t, d) from the main prediction; rewards (b, t); actions
    (b, t, na) int64; terminals (b, t) bool.  Returns dict(rewards=(mtp,), terminals=(), discrete_actions=(mtp,)) for what was given."""
    out = {}
    b, t = agent_embed.shape[:2]
    mtp = cfg.multi_token_pred_len
    lm = lens_to_mask(lens, t) if lens is not None else None                                # D4:7418-7423: frames past a trajectory's length
    lm_wo_last = lm[:, :-1] if lm is not None else None
    if rewards is not None:
        if cfg.reward_encoder_type == 'symexp_two_hot':
            two_hot = symexp_two_hot(rewards, cfg.reward_range, cfg.reward_num_bins)
        else:
            two_hot = hl_gauss_to_probs(cfg, rewards, cfg.reward_range, cfg.reward_num_bins)
        x = agent_embed[:, :-1]                                                            # each agent token predicts the NEXT rewards
        pred = torch.stack([rmsnorm(x, W['to_reward_pred.params.0'][i]) @ W['to_reward_pred.params.1'][i].t() for i in range(mtp)], dim=2)   # b t-1 mtp l
        tgt, mask = mtp_targets(two_hot[:, 1:], mtp)                                       # b t-1 mtp l
        losses = -(tgt * pred.log_softmax(dim=-1)).sum(dim=-1).masked_fill(~mask, 0.)
        out['rewards'] = losses[lm_wo_last].mean(dim=0) if lm is not None else losses.mean(dim=(0, 1))   # D4:7460-7463: the mean INCLUDES the mtp-masked zeros
    if terminals is not None and cfg.predict_terminals:
        pooled = latents[:, 1:].mean(dim=-2)
        logit = mlp(W, 'to_state_terminal_pred.0.', pooled, mlp_num_layers(cfg.terminal_mlp_depth), cfg.head_mlp_recipe).squeeze(-1)
        eps = 1. - cfg.gae_discount_factor
        tgt = terminals[:, 1:].float().clamp(min=eps, max=1. - eps)                        # D4:7481-7484
        tl = F.binary_cross_entropy_with_logits(logit, tgt, reduction='none')
        out['terminals'] = tl[lm_wo_last].mean() if lm is not None else tl.mean()
    if (actions is not None or cont_actions is not None) and t > 1:
        pe = policy_head(cfg, W, agent_embed)                                              # num_targets = t for (b, t, na) actions, D4:7553-7556
    if cont_actions is not None and t > 1:                                                 # Beta log-likelihood, D4:7566-7597 (stand-in parameterisation)
        padded = F.pad(cont_actions, (0, 0, 1, 0), value=0.)
        tgt, mask = mtp_targets(padded, mtp)
        tgt, mask = tgt[:, 1:].clamp(1e-5, 1. - 1e-5), mask[:, 1:]                         # soft_validate_range, D4:1439-1440
        per = []
        for i in range(mtp):
            params = torch.einsum('...d,ndt->...nt', pe, W['action_embedder.continuous_action_unembed'][:, i])
            nl = (-beta_log_prob(params, tgt[:, :, i], cfg.continuous_beta_param)).masked_fill(~mask[:, :, i, None], 0.)
            per.append(nl[lm].mean() if lm is not None else nl.mean())
        out['continuous_actions'] = torch.stack(per)
    if actions is not None and t > 1:
        padded = F.pad(actions, (0, 0, 1, 0), value=-1)                                    # sentinel, D4:7540
        tgt, mask = mtp_targets(padded, mtp)
        tgt, mask = tgt[:, 1:], mask[:, 1:]                                                # b t mtp na
        per = []
        for i in range(mtp):
            logits = pe @ W['action_embedder.discrete_action_unembed'][:, i].t()
            lp = discrete_log_probs(cfg, logits, tgt[:, :, i].clamp(min=0))
            nl = (-lp).masked_fill(~mask[:, :, i, None], 0.)
            per.append(nl[lm].mean() if lm is not None else nl.mean())                      # D4:7586-7590 (pred_len = t + 1: the full-length mask)
        out['discrete_actions'] = torch.stack(per)
    return out


def loss_normalize(state, name, loss, update, beta=0.95, eps=1e-6):
    """LossNormalizer.forward D4:645-669: loss / sqrt(exp_avg_sq) with the rms taken before the EMA update.  state: {name: exp_avg_sq}."""
    if state is None or name not in state:
        return loss
    rms = state[name].sqrt()
    if update:
        state[name] = torch.lerp(state[name], loss.detach().reshape(state[name].shape).square(), 1. - beta)
    return loss / rms.clamp(min=eps).reshape(loss.shape)


def dynamics_training_losses(cfg: Config, W, latents, noise, signal_levels, step_sizes_log2, shortcut_train, actions=None, rewards=None,
                             terminals=None, tasks=None, normalizers=None, update_loss_ema=True, lens=None, cont_actions=None):
    """Everything DynamicsWorldModel.forward returns in training for the supported subset (D4:6956-7743): flow, shortcut, rewards,
    terminals, discrete_actions and the total of D4:7708-7723 with unit loss weights (the reference defaults); `normalizers`: the
    LossNormalizer buffers by module name when `use_loss_normalization` (updated in place in the dict)."""
    times = signal_levels.float() / cfg.max_steps
    noised = noise.lerp(latents, times[:, :, None, None])
    _, agent_embed, _ = wm_forward(cfg, W, noised, signal_levels, step_sizes_log2, actions=actions, tasks=tasks, cont_actions=cont_actions)
    flow, short = dynamics_flow_losses(cfg, W, latents, noise, signal_levels, step_sizes_log2, shortcut_train, actions=actions, tasks=tasks, lens=lens,
                                       cont_actions=cont_actions)
    out = dict(flow=flow, shortcut=short, **dynamics_agent_losses(cfg, W, agent_embed, latents, rewards, actions, terminals, lens=lens,
                                                                  cont_actions=cont_actions))
    for key, name in (('flow', 'flow_loss_normalizer'), ('shortcut', 'shortcut_flow_loss_normalizer'), ('rewards', 'reward_loss_normalizer'),
                      ('terminals', 'state_terminal_loss_normalizer'), ('discrete_actions', 'discrete_actions_loss_normalizer'),
                      ('continuous_actions', 'continuous_actions_loss_normalizer')):
        if key in out:                                                                     # D4:7637-7654 (`normalizers`: {name: exp_avg_sq})
            out[key] = loss_normalize(normalizers, name, out[key], update_loss_ema)
    out['total'] = sum(v.sum() for v in out.values())
    return out


# ----------------------------------------------------------------------------- generate

def generate(cfg: Config, W, time_steps, *, num_steps=4, batch_size=1, noise, tasks=None,
             prompt_latents=None, prompt_discrete_actions=None, prompt_rewards=None, prompt_continuous_actions=None,
             cache: TrunkCache | None = None, use_time_cache=True, return_terminals=True,
             context_signal_noise=0.1, discrete_temperature=1., continuous_temperature=1., sample_actions=True):
    """DynamicsWorldModel.generate(return_rewards_per_frame, return_agent_actions,
    return_log_probs_and_values[, return_terminals]) D4:6308-6774, with every RNG draw injected:

      noise['latent'][f]   (b, n, dl)  normal   — D4:6475 for the f-th generated frame
      noise['context'][f]  (b, n, dl)  normal   — D4:6670
      noise['gumbel_u'][f] (b, A)      uniform  — MultiCategorical.sample
      noise['bern_u'][f]   (b,)        uniform  — torch.bernoulli (is_terminal = u < p)
      noise['beta'][f]     (b, nc, 2, rounds, 2) (normal, uniform) — Readout.sample_continuous (Beta as a ratio of gammas)

    Returns a dict with the Experience fields plus the final cache."""
    b = batch_size
    n, dl = cfg.num_latent_tokens, cfg.dim_latent
    na, nc = len(cfg.num_discrete_actions), cfg.num_continuous_actions
    step_size = cfg.max_steps // num_steps
    latents = prompt_latents.clone() if prompt_latents is not None else torch.zeros(b, 0, n, dl)
    ctx_noise = latents.clone()
    actions = prompt_discrete_actions.clone() if prompt_discrete_actions is not None else \
        torch.zeros(b, 0, na, dtype=torch.long)
    cont_actions = prompt_continuous_actions.clone() if prompt_continuous_actions is not None else torch.zeros(b, 0, nc)
    rewards = prompt_rewards.clone() if prompt_rewards is not None else torch.zeros(b, 0)
    log_probs, cont_log_probs, values, agent_embeds, policy_embeds = [], [], [], [], []
    terminals = torch.zeros(b, dtype=torch.bool)
    lens = torch.full((b,), time_steps)
    time_cache = cache

    def hist(t, cur):                                                              # D4:6515-6523
        if t.shape[1] == 0 or t.shape[-1] == 0:
            return None
        t = t[:, :cur]
        return F.pad(t, (0, 0, 0, cur - t.shape[1]), value=0) if t.shape[1] < cur else t

    f = 0
    while latents.shape[1] < time_steps:
        cur = latents.shape[1]
        x = noise['latent'][f].clone()[:, None]                          # b 1 n dl
        for step in range(num_steps + 1):
            last = step == num_steps
            sig_val = min(step * step_size, cfg.max_steps - 1)
            ctx = latents.lerp(ctx_noise, context_signal_noise)
            lat_in = torch.cat((ctx, x), dim=1)
            sig = torch.full((b, cur + 1), cfg.max_steps - 1, dtype=torch.long)
            sig[:, -1] = sig_val
            pred, agent_embed, next_cache = wm_forward(cfg, W, lat_in, sig, step_size, hist(actions, cur), tasks, time_cache,
                                                       cont_actions=hist(cont_actions, cur))
            if last:
                if use_time_cache:
                    time_cache = next_cache
                break
            pred = pred[:, -1:]
            t_ = sig_val / cfg.max_steps
            x = x + (pred - x) / (1. - t_) * (step_size / cfg.max_steps)          # D4:6567-6580

        one = agent_embed[:, -1:]                                                  # b 1 d
        rewards = torch.cat((rewards, reward_head(cfg, W, one)), dim=1)

        if return_terminals and cfg.predict_terminals:
            p = terminal_prob(cfg, W, x)                                           # b 1
            is_term = noise['bern_u'][f] < p[:, 0]
            just = is_term & ~terminals
            lens = torch.where(just, torch.full_like(lens, cur + 1), lens)
            terminals = terminals | is_term

        agent_embeds.append(one)
        if not sample_actions:                                  # return_agent_actions=False (D4:6625): no action conditioning
            latents = torch.cat((latents, x), dim=1)
            ctx_noise = torch.cat((ctx_noise, noise['context'][f][:, None]), dim=1)
            f += 1
            if return_terminals and cfg.predict_terminals and bool(terminals.all()):
                break
            continue
        pe = policy_head(cfg, W, one)
        policy_embeds.append(pe)
        if na > 0:
            logits = policy_logits(cfg, W, pe)                                     # b 1 A
            a = sample_discrete(cfg, logits, noise['gumbel_u'][f][:, None], discrete_temperature)
            actions = torch.cat((actions, a), dim=1)
            log_probs.append(discrete_log_probs(cfg, logits, a))
        if nc > 0:
            cp = policy_cont_params(cfg, W, pe)                                    # b 1 nc 2
            ca = sample_continuous(cp, noise['beta'][f][:, None], continuous_temperature, cfg.continuous_beta_param)
            cont_actions = torch.cat((cont_actions, ca), dim=1)
            cont_log_probs.append(beta_log_prob(cp, ca, cfg.continuous_beta_param))
        values.append(bins_to_scalar(cfg, value_head_bins(cfg, W, one), cfg.value_range, cfg.value_num_bins))

        latents = torch.cat((latents, x), dim=1)
        ctx_noise = torch.cat((ctx_noise, noise['context'][f][:, None]), dim=1)
        f += 1
        if return_terminals and cfg.predict_terminals and bool(terminals.all()):
            break

    latents = latents.clamp(-1., 1.)
    T = latents.shape[1]
    step_mask = torch.arange(T) < lens[:, None]
    if not sample_actions:
        return dict(latents=latents, agent_embed=torch.cat(agent_embeds, dim=1), rewards=rewards, lens=lens, terminals=terminals,
                    is_truncated=~terminals, episode_return=(rewards * step_mask.float()).sum(dim=-1), step_size=step_size,
                    cache=time_cache, frames_generated=f)
    policy_embeds = torch.cat(policy_embeds, dim=1)
    out = dict(
        latents=latents,
        agent_embed=torch.cat(agent_embeds, dim=1),
        rewards=rewards,
        values=torch.cat(values, dim=1),
        lens=lens,
        terminals=terminals,
        is_truncated=~terminals,
        episode_return=(rewards * step_mask.float()).sum(dim=-1),
        step_size=step_size,
        cache=time_cache,
        frames_generated=f,
    )
    if na > 0:
        out.update(actions=actions, log_probs=torch.cat(log_probs, dim=1), old_action_unembeds=policy_logits(cfg, W, policy_embeds))
    if nc > 0:
        out.update(actions_cont=cont_actions, log_probs_cont=torch.cat(cont_log_probs, dim=1),
                   old_cont_params=policy_cont_params(cfg, W, policy_embeds))
    return out


# ----------------------------------------------------------------------------- tokenizer decode

@dataclass
class TokenizerConfig:
    """VideoTokenizer constructor arguments the decode path reads (names follow D4:3686-3764); supported subset = the defaults:
    flow decoder with `decoder_flow_steps` Euler steps, no slot attention / causal conv / MOSS / aug conditioning / PoPE."""
    dim: int
    dim_latent: int
    patch_size: int
    image_height: int
    image_width: int
    num_latent_tokens: int = 64
    decoder_depth: int = 4
    time_block_every: int = 4
    attn_heads: int = 8
    attn_dim_head: int = 64
    attn_softclamp_value: float = 50.
    channels: int = 3
    decoder_pos_mlp_depth: int = 2
    decoder_flow_steps: int = 1
    head_mlp_recipe: str = 'pre_rms'
    encoder_depth: int = 4

    def encoder_trunk(self) -> Config:
        """The encoder's AxialSpaceTimeTransformer (D4:3912-3933): the num_latent_tokens latent tokens are the special tokens."""
        return Config(dim=self.dim, dim_latent=self.dim_latent, num_latent_tokens=self.num_latent_tokens, depth=self.encoder_depth,
                      time_block_every=self.time_block_every, attn_heads=self.attn_heads, attn_dim_head=self.attn_dim_head,
                      attn_softclamp_value=self.attn_softclamp_value)

    def trunk(self) -> Config:
        """The decoder's AxialSpaceTimeTransformer (D4:3582-3594: constructor defaults, i.e. ONE special token — the last latent
        token — with the final special cross-attention, attention pools, value residual and the final RMSNorm)."""
        return Config(dim=self.dim, dim_latent=self.dim_latent, num_latent_tokens=self.num_latent_tokens, depth=self.decoder_depth,
                      time_block_every=self.time_block_every, attn_heads=self.attn_heads, attn_dim_head=self.attn_dim_head,
                      attn_softclamp_value=50.)          # (the decoder does not forward attn_softclamp_value: D4:3582-3594)


def tokenizer_decode_step(tc: TokenizerConfig, W, latents, noised_video, time_index):
    """VideoTokenizer.decode_step D4:4137-4184 + VideoDecoderNetwork.forward D4:3599-3682.  latents (b, t, n, dl),
    noised_video (b, c, t, H, W) -> predicted clean video (b, c, t, H, W)."""
    b, t = latents.shape[:2]
    p, c = tc.patch_size, tc.channels
    nh, nw = tc.image_height // p, tc.image_width // p
    lat = latents @ W['latents_to_decoder.weight'].t() + W['time_embed.weight'][time_index]                      # D4:4151-4157
    # noised video -> patch tokens: 'b c t (h p1) (w p2) -> b t h w (p1 p2 c)', Linear, LayerNorm without bias    D4:3895-3899
    x = noised_video.reshape(b, c, t, nh, p, nw, p).permute(0, 2, 3, 5, 4, 6, 1).reshape(b, t, nh, nw, p * p * c)
    x = x @ W['noised_patch_to_tokens.1.weight'].t() + W['noised_patch_to_tokens.1.bias']
    x = layernorm(x, W['noised_patch_to_tokens.2.weight'], 0.)
    # positional embedding of the patch grid: MLP of the (row, column) coordinates in [-1, 1]                      D4:3617-3625
    gy, gx = torch.meshgrid(torch.linspace(-1., 1., nh), torch.linspace(-1., 1., nw), indexing='ij')
    pos = mlp(W, 'decoder.to_decoder_pos_emb.', torch.stack((gy, gx), dim=-1), mlp_num_layers(tc.decoder_pos_mlp_depth), tc.head_mlp_recipe)
    spatial = (pos + x).reshape(b, t, nh * nw, tc.dim)
    tokens = torch.cat((spatial, lat), dim=2)                                                                     # [patches | latents]  D4:3654
    tokens, _ = transformer(tc.trunk(), W, tokens, None, pre='decoder.transformer.')
    patches = tokens[:, :, :nh * nw] @ W['decoder.tokens_to_patch.0.weight'].t() + W['decoder.tokens_to_patch.0.bias']
    # 'b t h w (p1 p2 c) -> b c t (h p1) (w p2)'                                                                   D4:3556
    return patches.reshape(b, t, nh, nw, p, p, c).permute(0, 6, 1, 2, 4, 3, 5).reshape(b, c, t, nh * p, nw * p)


def tokenizer_tokenize(tc: TokenizerConfig, W, video):
    """VideoTokenizer.tokenize = forward(video, return_latents=True) in eval mode (no patch masking)  D4:4107-4113, 4239-4433.
    video (b, c, t, H, W) -> latents (b, t, n, dl) in (-1, 1)."""
    b, c, t, H, Wd = video.shape
    p = tc.patch_size
    nh, nw = H // p, Wd // p
    x = video.reshape(b, c, t, nh, p, nw, p).permute(0, 2, 3, 5, 4, 6, 1).reshape(b, t, nh * nw, p * p * c)     # D4:3838-3844
    x = x @ W['patch_to_tokens.1.weight'].t() + W['patch_to_tokens.1.bias']
    x = layernorm(x, W['patch_to_tokens.2.weight'], 0.)
    lat = W['latent_tokens'].expand(b, t, -1, -1)
    tokens = torch.cat((x, lat), dim=2)                                                                          # [patches | latents]  D4:4376
    tokens, _ = transformer(tc.encoder_trunk(), W, tokens, None, pre='encoder_transformer.', num_special=tc.num_latent_tokens)
    z = tokens[:, :, nh * nw:] @ W['encoded_to_latents.weight'].t()
    return z.tanh()


def tokenizer_decode(tc: TokenizerConfig, W, latents, noise):
    """VideoTokenizer.decode D4:4186-4237 with its one random draw injected: noise (b, c, t, H, W) normal (D4:4212)."""
    steps = tc.decoder_flow_steps
    video = noise
    for i in range(steps):
        time = i / steps
        pred = tokenizer_decode_step(tc, W, latents, video, i)
        video = video + (pred - video) / (1. - time) * (1. / steps)
    return video


# ----------------------------------------------------------------------------- learning

def calc_gae(rewards, values, masks, learn_masks, gamma, lam):
    """D4:1566-1600 (AssocScan reverse == plain reverse recurrence)."""
    masks = masks.float()
    v = F.pad(values, (0, 1), value=0.)
    v, v_next = v[..., :-1], v[..., 1:]
    delta = rewards + gamma * v_next * masks - v
    delta = delta.masked_fill(~learn_masks, 0.)
    gates = gamma * lam * masks
    gae = torch.zeros_like(delta)
    h = torch.zeros_like(delta[..., 0])
    for i in range(delta.shape[-1] - 1, -1, -1):
        h = gates[..., i] * h + delta[..., i]
        gae[..., i] = h
    return gae + v


def masked_mean(t, mask):
    return (t * mask).sum() / mask.sum() if mask.any() else (t * mask).sum()


def returns_and_advantage(cfg: Config, exp, normalize=True, eps=1e-6):
    """D4:5943-6024."""
    rewards, old_values, lens = exp['rewards'], exp['values'], exp['lens']
    T = rewards.shape[1]
    ar = torch.arange(T)
    gae_len_mask = ar < lens[:, None]
    rewards = rewards.masked_fill(~gae_len_mask, 0.)
    old_values = old_values.masked_fill(~gae_len_mask, 0.)
    learn_lens = lens - exp['is_truncated'].long()
    mask = ar < learn_lens[:, None]
    gae_masks = ar < (lens - 1).clamp(min=0)[:, None]
    term_seq = (ar == (lens - 1).clamp(min=0)[:, None]) & exp['terminals'][:, None]
    gae_masks = gae_masks & ~term_seq
    returns = calc_gae(rewards, old_values, gae_masks, mask, cfg.gae_discount_factor, cfg.gae_lambda)
    adv = returns - old_values
    if normalize:
        mean = masked_mean(adv, mask)
        var = masked_mean((adv - mean).pow(2), mask)
        adv = (adv - mean) / var.clamp(min=eps).sqrt()
    return returns, old_values, adv, mask


def learn_losses(cfg: Config, W, exp, objective='ppo', use_delight_gating=None, delight_temperature=None,
                 normalize_advantages=None, eps=1e-6, only_learn_policy_value_heads=True):
    """learn_from_experience with stored agent embeds  D4:5893-6305.
    Returns (total_policy_loss, value_loss); differentiable w.r.t. the head tensors in W — and, with
    only_learn_policy_value_heads=False (D4:6045-6075: the agent embeddings are recomputed by a forward WITH gradient over the stored
    latents and never detached), w.r.t. every world-model tensor in W.
    The three optional arguments default as the reference's do (D4:5905-5906, 6021)."""
    use_gate = cfg.use_delight_gating if use_delight_gating is None else use_delight_gating
    gate_temp = cfg.delight_temperature if delight_temperature is None else delight_temperature
    normalize = (objective != 'pmpo') if normalize_advantages is None else normalize_advantages
    returns, old_values, adv, mask = returns_and_advantage(cfg, exp, normalize=normalize, eps=eps)
    na, nc = len(cfg.num_discrete_actions), cfg.num_continuous_actions
    if exp.get('agent_embed') is None or not only_learn_policy_value_heads:
        # generate(store_agent_embed=False), or fine-tuning the whole world model: the agent embeddings are recomputed with ONE
        # parallel forward over the stored latents at the clean signal level, conditioned on the stored actions  D4:6045-6070
        lat = exp['latents']
        sig = torch.full(lat.shape[:2], cfg.max_steps - 1, dtype=torch.long)
        with torch.set_grad_enabled(not only_learn_policy_value_heads):
            _, agent_embeds, _ = wm_forward(cfg, W, lat, sig, exp['step_size'], exp.get('actions') if na > 0 else None, exp.get('tasks'), None,
                                            cont_actions=exp.get('actions_cont') if nc > 0 else None)
    else:
        agent_embeds = exp['agent_embed'].detach()
    Tn = agent_embeds.shape[1]

    pe = policy_head(cfg, W, agent_embeds)
    # discrete and continuous log-probs / entropies are concatenated over the action dimension and summed  D4:6090-6111
    lps, ents, olds = [], [], []
    if na > 0:
        logits = policy_logits(cfg, W, pe)
        dlp, dent = discrete_log_probs(cfg, logits, exp['actions'][:, -Tn:], with_entropy=True)
        lps.append(dlp); ents.append(dent); olds.append(exp['log_probs'])
    if nc > 0:
        cparams = policy_cont_params(cfg, W, pe)
        lps.append(beta_log_prob(cparams, exp['actions_cont'][:, -Tn:], cfg.continuous_beta_param)); ents.append(beta_entropy(cparams, cfg.continuous_beta_param)); olds.append(exp['log_probs_cont'])
    lp = torch.cat(lps, dim=-1).sum(dim=-1)
    ent = torch.cat(ents, dim=-1)
    old_lp = torch.cat(olds, dim=-1).sum(dim=-1)
    fmask = mask.float()

    gate = 1.
    if use_gate:
        gate = ((-lp * adv) / gate_temp).sigmoid().detach()

    if objective == 'ppo':
        ratio = (lp - old_lp).exp()
        clipped = ratio.clamp(1. - cfg.ppo_eps_clip, 1. + cfg.ppo_eps_clip)
        pl = -torch.min(ratio * adv, clipped * adv) * gate
        policy_loss = masked_mean(pl, mask)
    elif objective == 'spo':
        ratio = (lp - old_lp).exp()
        pl = -(ratio * adv - (adv.abs() * (ratio - 1.).square()) / (2 * cfg.ppo_eps_clip)) * gate
        policy_loss = masked_mean(pl, mask)
    elif objective == 'pmpo':
        pos = (adv >= 0.) & mask
        neg = (adv < 0.) & mask
        scaled = lp * gate * adv.tanh().abs()
        pos_loss = scaled[pos].sum() if pos.any() else 0.
        neg_loss = scaled[neg].sum() if neg.any() else 0.
        num = max(1., float(mask.sum()))
        policy_loss = -cfg.pmpo_pos_to_neg_weight * (pos_loss - neg_loss) / num
        if cfg.pmpo_kl_div_loss_weight > 0.:                                       # D4:6158-6182
            kl_loss = 0.
            if na > 0:
                new_l, old_l = logits, exp['old_action_unembeds']
                src, tgt = (old_l, new_l) if cfg.pmpo_reverse_kl else (new_l, old_l)
                kl, o = 0., 0
                for n in cfg.num_discrete_actions:
                    a, c = src[..., o:o + n].log_softmax(-1), tgt[..., o:o + n].log_softmax(-1)
                    kl = kl + (a.exp() * (a - c)).sum(dim=-1)
                    o += n
                kl_loss = kl_loss + masked_mean(kl, mask)
            if nc > 0:
                new_p, old_p = cparams, exp['old_cont_params']
                src, tgt = (old_p, new_p) if cfg.pmpo_reverse_kl else (new_p, old_p)
                kl_loss = kl_loss + masked_mean(beta_kl(src, tgt, cfg.continuous_beta_param).sum(dim=-1), mask)
            policy_loss = policy_loss + kl_loss * cfg.pmpo_kl_div_loss_weight
    else:
        raise ValueError(objective)

    entropy_loss = masked_mean(-ent.sum(dim=-1), mask)
    total_policy_loss = policy_loss + entropy_loss * cfg.policy_entropy_weight

    vbins = value_head_bins(cfg, W, agent_embeds)
    rbins = symexp_two_hot(returns, cfg.value_range, cfg.value_num_bins) if cfg.reward_encoder_type == 'symexp_two_hot' else \
        hl_gauss_to_probs(cfg, returns, cfg.value_range, cfg.value_num_bins)
    vl = -(rbins * vbins.log_softmax(dim=-1)).sum(dim=-1)
    value_loss = vl[mask].mean()
    return total_policy_loss, value_loss
