"""Host-side mirror of the reference `DynamicsWorldModel` for the imagination path.

Same constructor keyword names, same `generate` / `learn_from_experience` signatures and return
types as the reference (dreamer4/dreamer4.py:4661-4778, 5893-5904, 6308-6339); parameters carry
the reference's state_dict key names so checkpoints interchange as flat {key: tensor} (the keys under the three
normed head MLPs and `to_reward_pred` come from x-mlps-pytorch, which is not in the image: their recipe / spelling is
a descriptor, `head_mlp_recipe`, not a verified fact — DESIGN.md "Oracle").  All compute
is dispatched to the HIP engine through the C-ABI (include/d4hip.h); PyTorch only owns device
memory, the stream and (multi-GPU) the process group.  No CPU / eager fallback exists: on a
machine without the built extension or without a GPU these methods raise.

Configurations outside the supported subset raise NotImplementedError at
construction instead of being silently ignored.
"""
from __future__ import annotations

import ctypes as C
import math
from math import log2

import torch
from torch import nn

from dreamer4_amd import _lib
from dreamer4_amd.checkpoint import SaveLoad
from dreamer4_amd.experience import Actions, Experience


class _Node(nn.Module):
    """Plain container used to reproduce the reference's module tree (names only)."""


def _register(root: nn.Module, key: str, tensor: torch.Tensor, buffer=False, persistent=True):
    parts = key.split('.')
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, _Node())
        m = m._modules[p]
    if buffer:
        m.register_buffer(parts[-1], tensor, persistent=persistent)
    else:
        m.register_parameter(parts[-1], nn.Parameter(tensor))


def _linear_w(out_f, in_f):
    w = torch.empty(out_f, in_f)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    return w


def _linear_b(out_f, in_f):
    bound = 1 / math.sqrt(in_f) if in_f > 0 else 0
    return torch.empty(out_f).uniform_(-bound, bound)


def mlp_widths(dim_in, dim, dim_out, depth):
    """create_mlp(dim, depth, dim_in, dim_out) of x-mlps-pytorch: (dim_in, dim x (depth + 1), dim_out).
    ASSUMED (the package is not in the image) — see DESIGN.md 'Oracle / unpinned third-party pieces'."""
    return (dim_in, *((dim,) * (depth + 1)), dim_out)


# Layer recipe of the normed head MLPs (policy / value / terminal).  x-mlps-pytorch is absent from the image, so the recipe is ONE
# descriptor that the oracle shim, the restatement, this module tree and the HIP engine (csrc/engine.h `Mlp`) all honour; a real
# checkpoint that turns out to use the other recipe needs `head_mlp_recipe=...`, not a kernel change.
#   'pre_rms'     RMSNorm(d_in) -> Linear -> SiLU, no activation after the last Linear      keys layers.{i}.0.weight | layers.{i}.1.{weight,bias}
#   'post_layer'  Linear -> LayerNorm(d_out) -> SiLU, the last layer is a bare Linear       keys layers.{i}.0.{weight,bias} | layers.{i}.1.{weight,bias} ; last: layers.{i}.{weight,bias}
MLP_RECIPES = dict(pre_rms=0, post_layer=1)
# Link of the Beta policy head's raw outputs (discrete_continuous_embed_readout's BetaDist with unimodal=True, dreamer4.py:1172-1173):
# alpha = link(raw0) + 1, beta = link(raw1) + 1.  The package is absent from the image (pip offline), so — like the MLP recipe — the
# link is a descriptor every layer honours (kernels csrc/beta.h, oracle/restate.py, oracle/shim): the default is the stand-in's softplus;
# a checkpoint trained with the other published choice needs `continuous_beta_param='exp_p1'`, not a kernel change.
BETA_PARAMS = dict(softplus_p1=0, exp_p1=1)


def mlp_param_specs(recipe, widths):
    """[(key suffix, shape, kind)] of one normed MLP in state_dict order; kind in {'norm_w', 'norm_b', 'lin_w', 'lin_b'}."""
    out, n = [], len(widths) - 1
    for i, (a, b) in enumerate(zip(widths[:-1], widths[1:])):
        if recipe == 'pre_rms':
            out += [(f'layers.{i}.0.weight', (a,), 'norm_w'), (f'layers.{i}.1.weight', (b, a), 'lin_w'), (f'layers.{i}.1.bias', (b,), 'lin_b')]
        elif i < n - 1:
            out += [(f'layers.{i}.0.weight', (b, a), 'lin_w'), (f'layers.{i}.0.bias', (b,), 'lin_b'),
                    (f'layers.{i}.1.weight', (b,), 'norm_w'), (f'layers.{i}.1.bias', (b,), 'norm_b')]
        else:
            out += [(f'layers.{i}.weight', (b, a), 'lin_w'), (f'layers.{i}.bias', (b,), 'lin_b')]
    return out


class TimeCache:
    """Handle of the engine's time KV cache (reference: DynamicsIntermediates.main.next_kv_cache, token_count —
    dreamer4.py:3255-3265).  The reference hands out plain tensors, so an older cache can be passed again (K denoising
    evaluations against one cache, dreamer4.py:6510-6531).  Here the K/V of frame t live in slot t of the engine's ring and a
    handle stays usable without any copy as long as none of its slots has been rewritten since it was created (every call
    writes the slots from its own first new frame upwards); after that it needs the copy `kv()` made beforehand.
    `kv()` materialises the reference layout (time_layers, 2, B*S, heads, frames, attn_dim_head)."""

    def __init__(self, model, frames, batch, serial):
        self._model, self.frames, self.batch, self._serial = model, frames, batch, serial
        self._generation = model._engine_generation
        self._kv = None

    @property
    def token_count(self):
        return self.frames

    def kv(self):
        if self._kv is None:
            self._kv = self._model._export_cache(self)
        return self._kv


# buffers the engine never reads (key parity with the reference's state_dict only)
_NOT_BOUND = {'zero', 'ema_returns_mean', 'ema_returns_var', 'reward_loss_weight', 'terminal_loss_weight',
              'discrete_action_loss_weight', 'continuous_action_loss_weight'}
_LOSS_NORMALIZERS = ('flow_loss_normalizer', 'shortcut_flow_loss_normalizer', 'reward_loss_normalizer', 'state_terminal_loss_normalizer',
                     'discrete_actions_loss_normalizer', 'continuous_actions_loss_normalizer')

_UNSUPPORTED_DEFAULTS = dict(
    aux_image_encoder=None, num_agents=1, num_video_views=1, mot_temporal=False,
    dim_proprio=None, dim_state=None, dim_critic_state=None, critic_state_embedder=None,
    spatial_pre_encoder_depth=0, action_pre_encoder_depth=0, actor_depth=0, critic_depth=0,
    pred_orig_latent=True, use_time_rnn=False, add_reward_embed_to_agent_token=False,
    add_state_pred_head=False, agent_predicts_state=False, continuous_norm_stats=None, continuous_dist_type='beta',
    continuous_dist_kwargs={}, continuous_target_action_range=None,
    num_latent_genes=0, keep_reward_ema_stats=False, clip_values=False, time_attention_use_pope=False,
    latent_ar=False, identity_latents_to_spatial=False, has_aug_conditioning=False, ssl_lapo=False,
    ssl_tem=False, actor_spr=False, agent_policy_gradient_frac=1., agent_value_gradient_frac=1.,
    policy_head_mlp_activation='silu', value_head_mlp_activation='silu', state_terminal_pred_mlp_activation='silu',
)


class DynamicsWorldModel(SaveLoad, nn.Module):
    def __init__(
        self,
        dim,
        dim_latent,
        max_steps=64,
        num_register_tokens=8,
        num_spatial_tokens=4,
        num_latent_tokens=None,
        num_tasks=0,
        reward_encoder_type='hl_gauss',
        reward_encoder_kwargs: dict = dict(),
        value_encoder_kwargs: dict | None = None,
        depth=4,
        time_block_every=4,
        attn_kwargs: dict = dict(),
        transformer_kwargs: dict = dict(),
        attn_heads=8,
        attn_dim_head=64,
        attn_softclamp_value=50.,
        ff_kwargs: dict = dict(),
        num_discrete_actions: int | tuple = 0,
        num_continuous_actions=0,
        video_tokenizer=None,
        copy_video_tokenizer=True,
        multi_token_pred_len=8,
        value_head_mlp_depth=3,
        policy_head_mlp_depth=3,
        predict_terminals=True,
        predict_terminal_mlp_kwargs: dict = dict(depth=1),
        gae_discount_factor=0.997,
        gae_lambda=0.95,
        ppo_eps_clip=0.2,
        pmpo_pos_to_neg_weight=0.5,
        pmpo_reverse_kl=True,
        pmpo_kl_div_loss_weight=.3,
        use_delight_gating=True,
        delight_temperature=1.,
        normalize_advantages=None,
        policy_entropy_weight=.01,
        head_mlp_recipe='pre_rms',
        continuous_beta_param='softplus_p1',
        matmul_dtype='fp32',
        use_loss_normalization=False,
        latent_flow_loss_weight=1.,
        shortcut_loss_weight=1.,
        reward_loss_weight: float | list = 1.,
        terminal_loss_weight=1.,
        discrete_action_loss_weight: float | list = 1.,
        continuous_action_loss_weight: float | list = 1.,
        **kwargs,
    ):
        """`head_mlp_recipe` and `continuous_beta_param` are not reference arguments: they name the layer recipe of x_mlps_pytorch's normed
        MLP (MLP_RECIPES) and the link of discrete_continuous_embed_readout's Beta head (BETA_PARAMS), both third-party and unpinned."""
        config = dict(locals())
        super().__init__()
        self._record_config(config)
        if head_mlp_recipe not in MLP_RECIPES:
            raise ValueError(f'head_mlp_recipe must be one of {sorted(MLP_RECIPES)}')
        self.head_mlp_recipe = head_mlp_recipe
        if continuous_beta_param not in BETA_PARAMS:
            raise ValueError(f'continuous_beta_param must be one of {sorted(BETA_PARAMS)}')
        self.continuous_beta_param = continuous_beta_param
        # (not a reference argument) arithmetic of the trunk's GEMMs:
        #   'fp32' (default)  the reference's fp32; the wide projections (SiLU-GLU input, N >= 2048) run on the bf16 matrix cores with
        #                     every fp32 operand split exactly into three bf16 numbers and six products accumulated in fp32
        #                     (csrc/gemm_x3.hip: fp32 accuracy — measured error against float64 below the f32-input MFMA kernels'),
        #                     everything else on the f32-input MFMA
        #   'fp32_mfma'       every GEMM on the f32-input MFMA (the round-1 arithmetic)
        #   'fp32_fp16x2'     OPT-IN fast fp32-class mode (round 5, csrc/gemm_h2.hip): every operand row under an exact power-of-two scale as two
        #                     fp16 planes (a 23-bit image), three fp16 MFMA products per fp32 product, fp32 accumulate.  On this model's dot products
        #                     its error against float64 is BELOW the f32-input MFMA's (0.45x at K = 512) and the rollout is ~1.15x faster, but one
        #                     product carries up to 2^-21 relative error where fp32 has 2^-24 — it fails the sharpest of the three criteria the
        #                     default path is held to (profiles/r05_x3_products.txt), hence not the default
        #   'bf16'            bf16-rounded weights and activations on the bf16 MFMA, fp32 accumulation / norms / softmax / residual
        #                     stream (BASELINE config 5)
        if matmul_dtype not in ('fp32', 'fp32_mfma', 'fp32_fp16x2', 'bf16'):
            raise ValueError("matmul_dtype must be 'fp32', 'fp32_mfma', 'fp32_fp16x2' or 'bf16'")
        self.matmul_dtype = matmul_dtype
        self.use_loss_normalization = bool(use_loss_normalization)
        # loss weights of the training forward's total (dreamer4.py:4719-4725, 5257-5267, 7708-7723): two plain floats and four persistent
        # buffers of 1 or multi_token_pred_len elements — a checkpoint's values are loaded and used
        self.latent_flow_loss_weight, self.shortcut_loss_weight = float(latent_flow_loss_weight), float(shortcut_loss_weight)
        self._loss_weight_init = dict(reward_loss_weight=reward_loss_weight, terminal_loss_weight=terminal_loss_weight,
                                      discrete_action_loss_weight=discrete_action_loss_weight, continuous_action_loss_weight=continuous_action_loss_weight)
        for k, v in kwargs.items():
            if k not in _UNSUPPORTED_DEFAULTS:
                raise TypeError(f'unknown argument {k!r}')
            if v != _UNSUPPORTED_DEFAULTS[k]:
                raise NotImplementedError(f'{k}={v!r} is outside the supported imagination-path subset (SURVEY.md section 8)')
        if reward_encoder_type not in ('hl_gauss', 'symexp_two_hot'):
            raise AssertionError(f'unknown reward encoder type {reward_encoder_type}')
        self.reward_encoder_type = reward_encoder_type
        if attn_kwargs or transformer_kwargs or ff_kwargs:
            raise NotImplementedError('attn_kwargs / transformer_kwargs / ff_kwargs must be empty (reference defaults)')
        # the tokenizer decodes generated latents (generate(return_decoded_video=True), dreamer4.py:6694-6711) and tokenizes video prompts; like
        # the reference, its latent shape provides the defaults and it is a registered SUBMODULE (dreamer4.py:4787-4794: a frozen deep copy in
        # eval mode unless copy_video_tokenizer=False), so its weights travel in this model's state_dict under 'video_tokenizer.*' and a
        # checkpoint restores them.  The dynamics engine never binds those keys (the tokenizer owns its engines).
        if video_tokenizer is not None:
            if copy_video_tokenizer:
                import copy
                video_tokenizer = copy.deepcopy(video_tokenizer)
                video_tokenizer.requires_grad_(False)
            video_tokenizer = video_tokenizer.eval()
        self.video_tokenizer = video_tokenizer
        if video_tokenizer is not None:
            num_latent_tokens = num_latent_tokens if num_latent_tokens is not None else video_tokenizer.num_latent_tokens
            assert video_tokenizer.num_latent_tokens == num_latent_tokens and video_tokenizer.dim_latent == dim_latent, \
                'latent shape of the video tokenizer does not match the world model'
        if num_latent_tokens is None:
            raise AssertionError('`num_latent_tokens` must be set')
        if attn_dim_head not in (16, 32, 64):
            raise NotImplementedError('attn_dim_head must be 16, 32 or 64 (a head row lives in one CDNA wavefront)')
        assert dim % 2 == 0
        assert log2(max_steps).is_integer(), '`max_steps` must be a power of 2'

        nda = num_discrete_actions if isinstance(num_discrete_actions, (tuple, list)) else (num_discrete_actions,)
        nda = tuple(int(n) for n in nda if n > 0)
        value_encoder_kwargs = reward_encoder_kwargs if value_encoder_kwargs is None else value_encoder_kwargs

        def enc(kw):
            kw = dict(kw)
            rng = tuple(kw.pop('reward_range', (-20., 20.)))
            bins = kw.pop('num_bins', 255)
            ratio = kw.pop('sigma_to_bin_ratio', 2.)
            eps = kw.pop('eps', 1e-10)
            if kw or (reward_encoder_type == 'symexp_two_hot' and (ratio != 2. or eps != 1e-10)):
                raise NotImplementedError(f'reward/value encoder options {sorted(kw)} are not implemented')
            return rng, bins, ratio, eps

        (self.reward_range, self.reward_num_bins, _, _) = enc(reward_encoder_kwargs)
        (self.value_range, self.value_num_bins, self.hl_sigma_ratio, self.hl_eps) = enc(value_encoder_kwargs)

        self.dim, self.dim_latent, self.depth = dim, dim_latent, depth
        self.num_latent_tokens, self.latent_shape = num_latent_tokens, (num_latent_tokens, dim_latent)
        self.max_steps, self.num_register_tokens, self.num_spatial_tokens = max_steps, num_register_tokens, num_spatial_tokens
        self.num_tasks, self.time_block_every = num_tasks, time_block_every
        self.attn_heads, self.attn_dim_head, self.attn_softclamp_value = attn_heads, attn_dim_head, attn_softclamp_value
        self.num_discrete_actions = nda
        self.num_continuous_actions = int(num_continuous_actions)      # Beta policy head (the reference's default continuous_dist_type)
        self.multi_token_pred_len = multi_token_pred_len
        self.policy_head_mlp_depth, self.value_head_mlp_depth = policy_head_mlp_depth, value_head_mlp_depth
        self.terminal_mlp_depth = dict(predict_terminal_mlp_kwargs).get('depth', 1)
        self.predict_terminals = predict_terminals
        self.gae_discount_factor, self.gae_lambda, self.ppo_eps_clip = gae_discount_factor, gae_lambda, ppo_eps_clip
        self.pmpo_pos_to_neg_weight, self.pmpo_reverse_kl = pmpo_pos_to_neg_weight, pmpo_reverse_kl
        self.pmpo_kl_div_loss_weight = pmpo_kl_div_loss_weight
        self.use_delight_gating, self.delight_temperature = use_delight_gating, delight_temperature
        self.normalize_advantages, self.policy_entropy_weight = normalize_advantages, policy_entropy_weight
        self.pool_heads, self.pool_dim_head = 4, 64              # AttentionPool defaults, dreamer4.py:2147-2148
        # [flow | spatial | registers | action (only with an action space, dreamer4.py:7124-7130) | agent]
        self.tokens_per_frame = 1 + num_spatial_tokens + num_register_tokens + (1 if (len(nda) > 0 or self.num_continuous_actions > 0) else 0) + 1
        self.ff_inner = int(dim * 4 * 2 / 3)

        self._build_parameters()

        # engine state (created lazily on the parameters' device)
        self._engine = None
        self._engine_caps = None
        self._ws = None
        self._bound_sig = None
        self._trunk_version = None
        self._cache_serial = 0
        self._live_cache = None
        self._slot_serial = []           # per KV-ring slot: serial of the call that last wrote it
        self._engine_generation = 0      # bumped whenever the engine (and with it the ring) is rebuilt
        self._groups = {}

    # ------------------------------------------------------------------------------ parameters
    def _build_parameters(self):
        D, dl, h = self.dim, self.dim_latent, self.attn_heads
        dh = self.attn_dim_head
        hd = h * dh
        reg = lambda k, t: _register(self, k, t)

        def attn(pre, dim_q, dim_kv, heads, ctx_norm, mix, dh=dh):
            inner = heads * dh
            reg(pre + 'norm.weight', torch.ones(dim_q))
            if ctx_norm:
                reg(pre + 'norm_context.weight', torch.ones(dim_kv))
            reg(pre + 'to_q.weight', _linear_w(inner, dim_q))
            reg(pre + 'to_k.weight', _linear_w(inner, dim_kv))
            reg(pre + 'to_v.weight', _linear_w(inner, dim_kv))
            reg(pre + 'to_out.weight', _linear_w(dim_q, inner))
            reg(pre + 'to_gates.0.weight', _linear_w(heads, dim_q))
            reg(pre + 'k_heads_rmsnorm.gamma', torch.zeros(heads, dh))
            if mix:
                reg(pre + 'to_learned_value_residual_mix.0.weight', _linear_w(heads, dim_q))
                reg(pre + 'to_learned_value_residual_mix.0.bias', _linear_b(heads, dim_q))

        def ff(pre):
            reg(pre + 'norm.weight', torch.ones(D))
            reg(pre + 'proj_in.weight', _linear_w(2 * self.ff_inner, D))
            reg(pre + 'proj_in.bias', _linear_b(2 * self.ff_inner, D))
            reg(pre + 'proj_out.weight', _linear_w(D, self.ff_inner))
            reg(pre + 'proj_out.bias', _linear_b(D, self.ff_inner))

        def mlp(pre, widths):
            for key, shape, kind in mlp_param_specs(self.head_mlp_recipe, widths):
                if kind == 'norm_w':
                    reg(pre + key, torch.ones(shape))
                elif kind == 'norm_b':
                    reg(pre + key, torch.zeros(shape))
                elif kind == 'lin_w':
                    reg(pre + key, _linear_w(*shape))
                    fan_in = shape[1]
                else:
                    reg(pre + key, _linear_b(shape[0], fan_in))

        if self.num_spatial_tokens == self.num_latent_tokens:
            # one spatial token per latent token: Linear(dim_latent, dim) in, RMSNorm -> [Identity] -> Linear out   D4:4816-4834
            reg('latents_to_spatial_tokens.weight', _linear_w(D, dl))
            reg('latents_to_spatial_tokens.bias', _linear_b(D, dl))
            reg('to_latent_pred.0.weight', torch.ones(D))
        else:
            reg('latents_to_spatial_tokens.queries', torch.randn(self.num_spatial_tokens, D) * 1e-2)
            attn('latents_to_spatial_tokens.attn.', D, dl, h, True, False)
            reg('to_latent_pred.0.weight', torch.ones(D))
            reg('to_latent_pred.1.queries', torch.randn(self.num_latent_tokens, D) * 1e-2)
            attn('to_latent_pred.1.attn.', D, D, h, True, False)
        reg('to_latent_pred.2.weight', _linear_w(dl, D))
        reg('register_tokens', torch.randn(self.num_register_tokens, D) * 1e-2)
        reg('signal_levels_embed.weight', torch.randn(self.max_steps, D // 2))
        reg('step_size_embed.weight', torch.randn(int(log2(self.max_steps)), D // 2))
        reg('agent_learned_embed', torch.randn(1, D) * 1e-2)
        reg('action_learned_embed', torch.randn(1, D) * 1e-2)
        reg('reward_learned_embed', torch.randn(1, D) * 1e-2)          # unused on the supported path; kept for key parity
        reg('task_embed.weight', torch.randn(self.num_tasks, D))
        reg('latent_genes', torch.randn(0, D) * 1e-2)                   # num_latent_genes = 0 on the supported path; key parity (dreamer4.py:4946)
        mlp('policy_head.', mlp_widths(D, 4 * D, 4 * D, self.policy_head_mlp_depth))
        A = sum(self.num_discrete_actions)
        reg('action_embedder.discrete_action_unembed', torch.randn(A, self.multi_token_pred_len, 4 * D) * 1e-2)
        nc = self.num_continuous_actions
        reg('action_embedder.continuous_action_unembed', torch.randn(nc, self.multi_token_pred_len, 4 * D, 2) * 1e-2)
        reg('action_embedder.discrete_action_embed.weight', torch.randn(A, D))
        reg('action_embedder.continuous_action_embed.weight', torch.randn(nc, D))
        mtp = self.multi_token_pred_len
        reg('to_reward_pred.params.0', torch.ones(mtp, D))
        reg('to_reward_pred.params.1', torch.stack([_linear_w(self.reward_num_bins, D) for _ in range(mtp)]))
        if self.predict_terminals:
            mlp('to_state_terminal_pred.0.', mlp_widths(dl, 4 * dl, 1, self.terminal_mlp_depth))
        mlp('value_head.', mlp_widths(D, 4 * D, self.value_num_bins, self.value_head_mlp_depth))
        inv_freq = 1.0 / (10000. ** (torch.arange(0, dh, 2).float() / dh))
        _register(self, 'transformer.time_rotary.inv_freq', inv_freq, buffer=True)
        reg('transformer.to_value_residual.0.weight', torch.ones(D))
        reg('transformer.to_value_residual.1.weight', _linear_w(hd, D))
        for i in range(self.depth):
            attn(f'transformer.layers.{i}.2.fn.', D, D, h, False, True)
            ff(f'transformer.layers.{i}.3.fn.')
        for i in range(self.depth - 1):
            attn(f'transformer.attn_pools.{i}.fn.attn.', D, D, self.pool_heads, True, False, dh=self.pool_dim_head)
        attn('transformer.final_attn_pool.fn.attn.', D, D, self.pool_heads, True, False, dh=self.pool_dim_head)
        attn('transformer.final_special_cross_attn.fn.', D, D, h, True, True)
        ff('transformer.final_special_ff.fn.')
        # constructor-built buffers of the HL-Gauss encoders (hl_gauss_pytorch.HLGaussLoss.support / centers)
        for name, rng, bins in (('reward_encoder', self.reward_range, self.reward_num_bins),
                                ('value_encoder', self.value_range, self.value_num_bins)):
            if self.reward_encoder_type == 'symexp_two_hot':          # SymExpTwoHot.bin_values (a persistent buffer of the reference, dreamer4.py:958-963)
                values = torch.linspace(rng[0], rng[1], bins)
                _register(self, f'{name}.bin_values', values.sign() * (torch.exp(values.abs()) - 1.), buffer=True)
                continue
            support = torch.linspace(rng[0], rng[1], bins + 1).float()
            _register(self, f'{name}.support', support, buffer=True, persistent=False)
            _register(self, f'{name}.centers', (support[:-1] + support[1:]) / 2, buffer=True, persistent=False)
        _register(self, 'zero', torch.tensor(0.), buffer=True, persistent=False)
        # buffers of the reference's state_dict that the imagination path never reads (dreamer4.py:5245-5246, 5260-5263): registered
        # so that a reference checkpoint loads with strict=True
        for name, val in (('ema_returns_mean', 0.), ('ema_returns_var', 1.)):
            _register(self, name, torch.tensor(val), buffer=True)
        for name, val in self._loss_weight_init.items():
            w = torch.tensor(val, dtype=torch.float32)
            assert w.numel() in (1, self.multi_token_pred_len), f'{name}: 1 or multi_token_pred_len values (dreamer4.py:5265-5267)'
            _register(self, name, w, buffer=True)
        # LossNormalizer state of the training forward (dreamer4.py:629-669, 5250-5255): running mean of the squared loss per term
        if self.use_loss_normalization:
            mtp = self.multi_token_pred_len
            for name, n, on in (('flow_loss_normalizer', 1, True), ('shortcut_flow_loss_normalizer', 1, True), ('reward_loss_normalizer', mtp, True),
                                ('state_terminal_loss_normalizer', 1, self.predict_terminals),
                                # the reference creates both action normalizers whenever normalisation is on (`exists(0)` is true, dreamer4.py:5254-5255)
                                ('discrete_actions_loss_normalizer', mtp, True),
                                ('continuous_actions_loss_normalizer', mtp, True)):
                if on:
                    _register(self, name + '.exp_avg_sq', torch.ones(n), buffer=True)

    def _normalize_loss(self, name, loss, update_ema, beta=0.95, eps=1e-6):
        """LossNormalizer.forward (dreamer4.py:645-669): divide by the root of the running mean square (taken BEFORE this call's update)."""
        if not self.use_loss_normalization or not hasattr(self, name):
            return loss
        buf = getattr(self, name).exp_avg_sq
        rms = buf.sqrt()
        if update_ema:
            with torch.no_grad():
                buf.lerp_(loss.detach().reshape(buf.shape).square(), 1. - beta)
        return loss / rms.clamp(min=eps).reshape(loss.shape)

    @property
    def device(self):
        return self.zero.device

    def head_mlp_output_linear(self, head):
        """(weight, bias) of the last Linear of a normed head MLP ('policy_head', 'value_head', 'to_state_terminal_pred.0'),
        whatever the recipe names it."""
        widths_n = dict(policy_head=self.policy_head_mlp_depth, value_head=self.value_head_mlp_depth)
        last = (widths_n[head] if head in widths_n else self.terminal_mlp_depth) + 1
        pre = f'{head}.layers.{last}.' + ('1.' if self.head_mlp_recipe == 'pre_rms' else '')
        params = dict(self.named_parameters())
        return params[pre + 'weight'], params[pre + 'bias']

    # x_mlps_pytorch.Ensemble's parameter naming is as unverifiable here as the MLP recipe: accept the plausible spellings of the
    # two stacked tensors of `to_reward_pred` (RMSNorm weight, Linear weight) when loading a checkpoint
    _ENSEMBLE_ALIASES = {'to_reward_pred.params.0': ('to_reward_pred.param_values.0', 'to_reward_pred.params.0_weight', 'to_reward_pred.params.0.weight',
                                                     'to_reward_pred.ensemble_params.0.weight'),
                         'to_reward_pred.params.1': ('to_reward_pred.param_values.1', 'to_reward_pred.params.1_weight', 'to_reward_pred.params.1.weight',
                                                     'to_reward_pred.ensemble_params.1.weight')}

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for ours, aliases in self._ENSEMBLE_ALIASES.items():
            if prefix + ours not in state_dict:
                for a in aliases:
                    if prefix + a in state_dict:
                        state_dict[prefix + ours] = state_dict.pop(prefix + a)
                        break
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def load_state_dict(self, state_dict, strict=True, allow_missing_tokenizer=False, **kwargs):
        """The nested tokenizer is a registered submodule as in the reference (dreamer4.py:4787-4794), so a strict load RAISES on a checkpoint without
        `video_tokenizer.*` keys — a checkpoint saved with video_tokenizer=None, or one with its tokenizer keys stripped, must not pass silently and leave
        unrelated tokenizer weights in place.  `allow_missing_tokenizer=True` is the explicit opt-in for checkpoints written before round 4 (which kept the
        tokenizer outside the module tree): only when NO tokenizer key is present, the tokenizer this model was constructed with keeps its weights, every other
        key is still checked, a warning says so and `self.tokenizer_restored` records it (True after a load that carried the tokenizer)."""
        has_tok = any(k.startswith('video_tokenizer.') for k in state_dict)
        if self.video_tokenizer is not None and not has_tok and allow_missing_tokenizer:
            import warnings
            warnings.warn('load_state_dict: the checkpoint holds no video_tokenizer.* keys - the tokenizer of this model keeps the weights it was constructed '
                          'with (allow_missing_tokenizer=True)', stacklevel=2)
            state_dict = dict(state_dict)
            state_dict.update({'video_tokenizer.' + k: v for k, v in self.video_tokenizer.state_dict().items()})
        out = super().load_state_dict(state_dict, strict=strict, **kwargs)
        self.tokenizer_restored = bool(has_tok) if self.video_tokenizer is not None else None
        return out

    def policy_head_parameters(self):
        """dreamer4.py:5343-5355"""
        return [*self.policy_head.parameters(), self.action_embedder.discrete_action_unembed,
                self.action_embedder.continuous_action_unembed]

    def value_head_parameters(self):
        """dreamer4.py:5357-5363"""
        return list(self.value_head.parameters())

    # ------------------------------------------------------------------------------ engine plumbing
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _make_config(self, caps):
        c = _lib.Config()
        c.dim, c.dim_latent, c.num_latent_tokens, c.depth = self.dim, self.dim_latent, self.num_latent_tokens, self.depth
        c.time_block_every, c.attn_heads, c.attn_dim_head = self.time_block_every, self.attn_heads, self.attn_dim_head
        c.attn_softclamp_value = self.attn_softclamp_value
        c.num_spatial_tokens, c.num_register_tokens = self.num_spatial_tokens, self.num_register_tokens
        c.max_steps, c.num_tasks = self.max_steps, self.num_tasks
        c.num_discrete_action_types = len(self.num_discrete_actions)
        for i, n in enumerate(self.num_discrete_actions):
            c.num_discrete_actions[i] = n
        c.num_continuous_actions = self.num_continuous_actions
        c.multi_token_pred_len = self.multi_token_pred_len
        c.policy_head_mlp_depth, c.value_head_mlp_depth = self.policy_head_mlp_depth, self.value_head_mlp_depth
        c.terminal_mlp_depth, c.predict_terminals = self.terminal_mlp_depth, int(self.predict_terminals)
        c.reward_num_bins, c.value_num_bins = self.reward_num_bins, self.value_num_bins
        c.head_mlp_recipe = MLP_RECIPES[self.head_mlp_recipe]
        c.continuous_beta_param = BETA_PARAMS[self.continuous_beta_param]
        c.reward_encoder_type = int(self.reward_encoder_type == 'symexp_two_hot')
        c.matmul_bf16 = {'fp32': 2, 'fp32_mfma': 0, 'bf16': 1, 'fp32_fp16x2': 3}[self.matmul_dtype]
        c.pool_heads, c.pool_dim_head = self.pool_heads, self.pool_dim_head
        c.gae_discount_factor, c.gae_lambda, c.ppo_eps_clip = self.gae_discount_factor, self.gae_lambda, self.ppo_eps_clip
        c.policy_entropy_weight = self.policy_entropy_weight
        c.use_delight_gating, c.delight_temperature = int(self.use_delight_gating), self.delight_temperature
        c.pmpo_pos_to_neg_weight, c.pmpo_kl_div_loss_weight = self.pmpo_pos_to_neg_weight, self.pmpo_kl_div_loss_weight
        c.pmpo_reverse_kl = int(self.pmpo_reverse_kl)
        c.hl_gauss_sigma_to_bin_ratio, c.hl_gauss_eps = self.hl_sigma_ratio, self.hl_eps
        c.value_min, c.value_max = self.value_range
        c.max_batch, c.max_frames, c.max_parallel_frames, c.max_learn_rows = caps
        return c

    def _ensure_engine(self, batch=1, frames=1, parallel=1, learn_rows=0):
        if self.device.type != 'cuda':
            raise _lib.D4Error('the imagination path runs only on an MI355X (HIP) device: move the model with .cuda(); '
                               'there is no CPU fallback')
        lib = _lib.load()
        caps = self._engine_caps
        if caps is None or batch > caps[0] or frames > caps[1] or parallel > caps[2] or learn_rows > caps[3]:
            keep = None
            if self._engine is not None and lib.d4_engine_cache_frames(self._engine) > 0 and self._live_cache is not None:
                keep = (self._live_cache, self._live_cache.kv())
            old = caps or (0, 0, 0, 0)
            # the KV-cache capacity grows geometrically: the env-wrapper pattern (one more frame per call) would otherwise
            # rebuild the engine (workspace + weight preparation) on every call
            if frames > old[1]:
                frames = max(16, 1 << (frames - 1).bit_length())
            caps = (max(batch, old[0]), max(frames, old[1]), max(parallel, old[2]), max(learn_rows, old[3]))
            if self._engine is not None:
                lib.d4_engine_destroy(self._engine)
            eng = C.c_void_p()
            cfg = self._make_config(caps)
            _lib.check(lib.d4_engine_create(C.byref(cfg), C.byref(eng)))
            self._engine, self._engine_caps = eng, caps
            self._engine_generation += 1
            self._slot_serial = [0] * max(caps[1], caps[2])
            nbytes = lib.d4_engine_workspace_bytes(eng)
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            base = self._ws.data_ptr()
            self._ws_off = (-base) % 256
            _lib.check(lib.d4_engine_set_workspace(eng, C.c_void_p(base + self._ws_off), nbytes))
            self._bound_sig = None
            self._trunk_version = None
            self._bind_cache = None
            if keep is not None:
                tc, kv = keep
                _lib.check(lib.d4_engine_cache_import(eng, _lib.ptr(kv), tc.batch, tc.frames, self._stream()))
                tc._generation = self._engine_generation             # the live handle moves to the new ring
                self._mark_slots(0, tc.frames, tc._serial)
        self._bind()
        return self._engine

    def _flatten_group(self, name, params):
        """Make a head's parameters views of one flat buffer (and allocate its flat gradient), so the
        optimiser step and the RCCL all-reduce each touch a single contiguous tensor."""
        params = [p for p in params if p.numel() > 0]
        total = sum(p.numel() for p in params)
        g = self._groups.get(name)
        ok = g is not None and g['flat'].device == self.device and g['flat'].numel() == total
        if ok:
            off = 0
            for p in params:
                if p.data_ptr() != g['flat'].data_ptr() + 4 * off:
                    ok = False
                    break
                off += p.numel()
        if ok:
            return g
        flat = torch.empty(total, dtype=torch.float32, device=self.device)
        grad = torch.zeros(total, dtype=torch.float32, device=self.device)
        off = 0
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            off += n
        g = dict(flat=flat, grad=grad, params=params)
        self._groups[name] = g
        return g

    def _bind(self):
        lib = _lib.load()
        # fast path (every call): nothing moved and no trunk weight was written since the last bind/prepare
        cached = getattr(self, '_bind_cache', None)
        if cached is not None and self._bound_sig is not None:
            tensors, trunk = cached
            if all(t.data_ptr() == p for t, p in tensors) and tuple(t._version for t in trunk) == self._trunk_version:
                return
        groups = dict(policy=self._flatten_group('policy', self.policy_head_parameters()),
                      value=self._flatten_group('value', self.value_head_parameters()))
        grads = {}
        for g in groups.values():
            off = 0
            for p in g['params']:
                grads[id(p)] = g['grad'][off:off + p.numel()]
                off += p.numel()
        tensors = {k: v for k, v in self.named_parameters() if not k.startswith('video_tokenizer.')}
        tensors.update({k: v for k, v in self.named_buffers() if k not in _NOT_BOUND and not k.startswith(_LOSS_NORMALIZERS + ('video_tokenizer.',))})
        sig = tuple((k, t.data_ptr(), t.numel()) for k, t in tensors.items())
        if sig != self._bound_sig:
            for k, t in tensors.items():
                if t.numel() == 0:
                    continue
                assert t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device, k
                gr = grads.get(id(t))
                _lib.check(lib.d4_engine_bind(self._engine, k.encode(), _lib.ptr(t), _lib.ptr(gr), t.numel()))
            self._bound_sig = sig
            self._trunk_version = None
        head_ids = {id(p) for g in groups.values() for p in g['params']}
        trunk = [t for t in tensors.values() if id(t) not in head_ids]
        ver = tuple(t._version for t in trunk)
        if ver != self._trunk_version:
            _lib.check(lib.d4_engine_prepare(self._engine, self._stream()))
            self._trunk_version = ver
        self._bind_cache = ([(t, t.data_ptr()) for t in tensors.values()], trunk)

    def _mark_slots(self, lo, hi, serial):
        if hi > len(self._slot_serial):
            self._slot_serial.extend([0] * (hi - len(self._slot_serial)))
        for i in range(lo, hi):
            self._slot_serial[i] = serial

    def _cache_in_engine(self, tc: TimeCache):
        """True while every ring slot the handle covers still holds what it held when the handle was created."""
        return (tc._generation == self._engine_generation and tc.frames <= len(self._slot_serial)
                and all(sv <= tc._serial for sv in self._slot_serial[:tc.frames]))

    def _export_cache(self, tc: TimeCache):
        lib = _lib.load()
        if not self._cache_in_engine(tc):
            raise _lib.D4Error('this TimeCache is stale (later calls have rewritten its slots of the engine cache); '
                               'call .kv() before issuing them to keep a copy')
        lt = sum(1 for i in range(self.depth) if (i + 1) % self.time_block_every == 0)
        out = torch.empty(lt, 2, tc.batch * self.tokens_per_frame, self.attn_heads, tc.frames, self.attn_dim_head, device=self.device)
        now = lib.d4_engine_cache_frames(self._engine)
        _lib.check(lib.d4_engine_cache_reset(self._engine, tc.frames))
        _lib.check(lib.d4_engine_cache_export(self._engine, _lib.ptr(out), tc.batch, self._stream()))
        _lib.check(lib.d4_engine_cache_reset(self._engine, now))
        return out

    def _adopt_cache(self, time_cache, batch):
        lib = _lib.load()
        if time_cache is None:
            _lib.check(lib.d4_engine_cache_reset(self._engine, 0))
            self._live_cache = None
            return 0
        assert isinstance(time_cache, TimeCache), 'time_cache must come from generate(..., return_time_cache=True)'
        assert time_cache.batch == batch, 'time_cache batch size mismatch'
        if self._cache_in_engine(time_cache):
            _lib.check(lib.d4_engine_cache_reset(self._engine, time_cache.frames))       # rewind / fast-forward the frame counter only
        else:
            if time_cache._kv is None:
                raise _lib.D4Error('stale TimeCache without a materialised copy (.kv())')
            _lib.check(lib.d4_engine_cache_import(self._engine, _lib.ptr(time_cache._kv), batch, time_cache.frames, self._stream()))
            time_cache._generation = self._engine_generation
            self._cache_serial += 1
            time_cache._serial = self._cache_serial
            self._mark_slots(0, time_cache.frames, time_cache._serial)
        self._live_cache = time_cache
        return time_cache.frames

    def invalidate_prepared(self):
        """Force the fused / gamma-folded weight images to be rebuilt on the next call.  Needed only after writing a trunk weight
        in a way autograd's version counter cannot see (`p.data.copy_(...)`, `dist.broadcast(p.data)`, raw pointers); ordinary
        in-place updates (`p.copy_()` under no_grad, optimiser steps, `load_state_dict`) are detected automatically."""
        self._trunk_version = None

    # ------------------------------------------------------------------------------ forward (inference branch / training branch)
    def forward(self, *, latents, signal_levels=None, step_sizes=None, discrete_actions=None, continuous_actions=None, tasks=None, time_cache=None,
                latent_is_noised=None, return_pred_only=None, return_intermediates=True, commit_cache=True, **kwargs):
        """DynamicsWorldModel.forward (dreamer4.py:6792-7743).
        Inference branch (`signal_levels` / `step_sizes` given, latents already noised): returns (pred_flow, (agent_embed, next_time_cache))
        from the engine.  Training branch (neither given): samples the shortcut coin, step sizes, signal levels and noise as the
        reference does (dreamer4.py:6956-7003) and returns the total loss of dreamer4.py:7708-7723 (`return_all_losses=True`: `(total,
        WorldModelLosses(flow, shortcut, rewards, terminals, discrete_actions, continuous_actions))`): flow + shortcut, and — when `rewards` / `terminals` /
        `discrete_actions` are given — the multi-token-prediction reward, terminal and behaviour-cloning losses (dreamer4.py:7432-7598);
        differentiable through the HIP trunk blocks (dreamer4_amd/trunk_ops.py); `lens` masks frames past each trajectory's length out of
        every term.  Not implemented: proprio.  `use_loss_normalization=True` (constructor) applies the reference's LossNormalizer per term."""
        if signal_levels is None and step_sizes is None:
            return self._training_forward(latents, discrete_actions, continuous_actions, tasks, **kwargs)
        with torch.no_grad():
            return self._inference_forward(latents, signal_levels, step_sizes, discrete_actions, continuous_actions, tasks, time_cache,
                                           latent_is_noised, return_pred_only, commit_cache, kwargs)

    def _inference_forward(self, latents, signal_levels, step_sizes, discrete_actions, continuous_actions, tasks, time_cache, latent_is_noised,
                           return_pred_only, commit_cache, kwargs):
        latent_is_noised = True if latent_is_noised is None else latent_is_noised
        return_pred_only = True if return_pred_only is None else return_pred_only
        if kwargs:
            raise NotImplementedError(f'forward(): arguments {sorted(kwargs)} are outside the inference branch')
        if not (latent_is_noised and return_pred_only):
            raise NotImplementedError('with signal_levels / step_sizes only forward(latent_is_noised=True, return_pred_only=True) is implemented')
        B, T = latents.shape[:2]
        step = int(step_sizes if not torch.is_tensor(step_sizes) else step_sizes.flatten()[0].item())
        cached = time_cache.frames if time_cache is not None else 0
        self._ensure_engine(batch=B, frames=cached + T, parallel=T)
        lib = _lib.load()
        self._adopt_cache(time_cache, B)
        dev = self.device
        if isinstance(signal_levels, int):
            signal_levels = torch.full((B, T), signal_levels)
        sig = signal_levels.to(dev).expand(B, T).to(torch.int32).contiguous() if signal_levels.ndim == 2 else \
            signal_levels.to(dev).reshape(-1, 1).expand(B, T).to(torch.int32).contiguous()
        na, nc = len(self.num_discrete_actions), self.num_continuous_actions

        def pair(a, pad):
            """action token of frame t = the action of frame t-1 (dreamer4.py:7103-7115); frame 0 has none (`pad`)"""
            if a is None or a.shape[1] == 0:
                return None
            a = a.to(dev)
            a = a[..., None] if a.ndim == 2 else a
            if time_cache is not None and T == 1 and a.shape[1] == 1:
                return a.contiguous()                                   # sequential step: already paired
            if a.shape[1] == T:
                a = a[:, :-1]
            assert a.shape[1] == T - 1
            return torch.cat((torch.full((B, 1, a.shape[-1]), pad, dtype=a.dtype, device=dev), a), dim=1).contiguous()

        prev = pair(discrete_actions.long() if discrete_actions is not None else None, -1)
        prev_c = pair(continuous_actions.float() if continuous_actions is not None else None, float('nan') if na == 0 else 0.)
        assert (prev is None) == (prev_c is None) or na == 0 or nc == 0, 'pass both discrete and continuous actions, or neither'
        tk = tasks.to(dev).long().contiguous() if tasks is not None else None
        pred = torch.empty(B, T, *self.latent_shape, device=dev)
        agent = torch.empty(B, T, self.dim, device=dev)
        lat = latents.to(dev).float().reshape(B, T, *self.latent_shape).contiguous()
        _lib.check(lib.d4_wm_forward(self._engine, _lib.ptr(lat), _lib.ptr(sig), step, _lib.ptr(prev), _lib.ptr(prev_c), _lib.ptr(tk), B, T,
                                     int(time_cache is not None), int(commit_cache), _lib.ptr(pred), _lib.ptr(agent), self._stream()))
        self._cache_serial += 1
        self._mark_slots(cached, cached + T, self._cache_serial)      # every evaluation writes its new frames' K/V (scratch unless committed)
        if not commit_cache:
            return pred, (agent, time_cache)                           # the cache passed in is unchanged and still the live one
        tc = TimeCache(self, lib.d4_engine_cache_frames(self._engine), B, self._cache_serial)
        self._live_cache = tc
        return pred, (agent, tc)

    def _shortcut_coin(self, generator, prob):
        """The shortcut / plain-flow coin of a training step (dreamer4.py:345-346, 6965: `rand(1).item() < prob`, a HOST draw in the reference).
        Drawn on the host here too — a device draw would make every training step wait for the device — from the global CPU generator, or,
        when the caller passes a (device) generator, from a CPU companion seeded with its initial seed (deterministic per seed).
        Consequences worth knowing: the coin does NOT advance the passed generator (the randint / randn draws that follow are not shifted by
        it); the companion restarts whenever the generator object or its seed changes, or the generator is found freshly (re-)seeded — so
        re-seeding the same object with the same seed replays the same coin sequence; `draws=dict(shortcut_train=...)` injects the coin outright."""
        if generator is None:
            return bool(torch.rand(1).item() < prob)
        pair = getattr(self, '_coin_generator', None)
        seed = generator.initial_seed()
        fresh = pair is not None and pair[0] is generator and pair[2] == seed and torch.equal(generator.get_state(), pair[3])
        if pair is None or pair[0] is not generator or pair[2] != seed or fresh:
            # (a generator whose state equals the state right after manual_seed(seed) was just (re-)seeded: the replay starts the coin sequence over)
            ref_state = torch.Generator(device=generator.device).manual_seed(seed).get_state()
            pair = (generator, torch.Generator().manual_seed(seed), seed, ref_state)
            object.__setattr__(self, '_coin_generator', pair)
        return bool(torch.rand(1, generator=pair[1]).item() < prob)

    def _training_forward(self, latents, discrete_actions, continuous_actions, tasks, *, return_all_losses=False, seed=None, generator=None,
                          add_autoregressive_action_loss=True, prob_shortcut_train=None, draws=None, rewards=None, terminals=None,
                          update_loss_ema=None, lens=None, **kwargs):
        """Training branch: flow loss (x-space, ramp weight) + shortcut consistency loss (dreamer4.py:6956-7003, 7335-7431) + the
        agent-token losses (dreamer4.py:7432-7598) + total (dreamer4.py:7708-7723).
        `draws` = dict(shortcut_train, step_sizes_log2, signal_levels, noise) injects the random draws (parity runs); otherwise they
        come from `generator` (or a generator seeded with `seed`, as the reference's `seed=`)."""
        from dreamer4_amd import trunk_ops
        unsupported = {k: v for k, v in kwargs.items() if v is not None}
        if unsupported:
            raise NotImplementedError(f'training forward: {sorted(unsupported)} is not implemented (proprio / video / aug / genes are outside the built slice)')
        if self.reward_encoder_type != 'hl_gauss' and rewards is not None:
            raise NotImplementedError('training forward: reward loss with the symexp_two_hot encoder is not implemented')
        dev = self.device
        lat = latents.to(dev).float()
        if lat.ndim == 5:
            assert lat.shape[2] == 1
            lat = lat[:, :, 0]
        B, T = lat.shape[:2]
        assert lat.shape[2:] == tuple(self.latent_shape), f'latents must have shape {self.latent_shape}'
        n_log2 = int(log2(self.max_steps))
        if draws is None:
            g = generator
            if g is None and seed is not None:
                g = torch.Generator(device=dev).manual_seed(seed)
            prob = (1. - n_log2 ** -1.) if prob_shortcut_train is None else prob_shortcut_train           # dreamer4.py:4898
            shortcut = self._shortcut_coin(g, prob)
            if shortcut:                                                                                    # dreamer4.py:6967-6974, eq. (4)
                step_log2 = torch.randint(1, n_log2, (B,), device=dev, generator=g)
                nss = (2 ** step_log2)[:, None]
                sig = torch.randint(0, self.max_steps, (B, T), device=dev, generator=g) // nss * nss
            else:
                step_log2 = torch.zeros(B, dtype=torch.long, device=dev)
                sig = torch.randint(0, self.max_steps, (B, T), device=dev, generator=g)
            noise = torch.randn(lat.shape, device=dev, generator=g)
        else:
            shortcut, step_log2, sig, noise = bool(draws['shortcut_train']), draws['step_sizes_log2'].to(dev).long(), draws['signal_levels'].to(dev).long(), draws['noise'].to(dev).float()
        W = dict(self.named_parameters())
        W.update({k: v for k, v in self.named_buffers() if k.endswith('inv_freq')})
        is_time = [(i + 1) % self.time_block_every == 0 for i in range(self.depth)]
        flow, short, agent_embed = trunk_ops.dynamics_flow_losses(
            W, lat, noise, sig, step_log2, shortcut, max_steps=self.max_steps, return_agent_embed=True,
            lens=lens.to(dev) if lens is not None else None, is_time=is_time, num_spatial_tokens=self.num_spatial_tokens,
            num_register_tokens=self.num_register_tokens, num_discrete_actions=tuple(self.num_discrete_actions),
            discrete_actions=discrete_actions.to(dev).long() if discrete_actions is not None else None,
            continuous_actions=continuous_actions.to(dev).float() if continuous_actions is not None else None,
            tasks=tasks.to(dev).long() if tasks is not None else None, softclamp_value=self.attn_softclamp_value)
        rew = rewards.to(dev).float() if rewards is not None else None
        if rew is not None and rew.shape[1] == T - 1:
            rew = torch.nn.functional.pad(rew, (1, 0), value=0.)                                           # dreamer4.py:6905-6907
        term = terminals.to(dev) if (terminals is not None and self.predict_terminals) else None
        if term is not None:
            assert term.ndim == 2, 'terminals must be (batch, time) or (batch, time - 1)'
            if term.shape[1] == T - 1:
                term = torch.nn.functional.pad(term, (1, 0), value=False)
        da = discrete_actions.to(dev).long() if (discrete_actions is not None and add_autoregressive_action_loss) else None
        if da is not None and da.ndim == 2:
            da = da[..., None]
        ca = continuous_actions.to(dev).float() if (continuous_actions is not None and add_autoregressive_action_loss) else None
        agent = trunk_ops.dynamics_agent_losses(
            W, agent_embed, lat, multi_token_pred_len=self.multi_token_pred_len, num_discrete_actions=tuple(self.num_discrete_actions),
            reward_range=self.reward_range, reward_num_bins=self.reward_num_bins, policy_head_mlp_depth=self.policy_head_mlp_depth,
            terminal_mlp_depth=self.terminal_mlp_depth, head_mlp_recipe=self.head_mlp_recipe, continuous_beta_param=self.continuous_beta_param, gae_discount_factor=self.gae_discount_factor,
            hl_sigma_ratio=self.hl_sigma_ratio, hl_eps=self.hl_eps, rewards=rew, discrete_actions=da, terminals=term,
            lens=lens.to(dev) if lens is not None else None, continuous_actions=ca)
        upd = self.training if update_loss_ema is None else bool(update_loss_ema)                           # dreamer4.py:7637-7654
        flow = self._normalize_loss('flow_loss_normalizer', flow, upd)
        short = self._normalize_loss('shortcut_flow_loss_normalizer', short, upd)
        for key, name in (('rewards', 'reward_loss_normalizer'), ('terminals', 'state_terminal_loss_normalizer'),
                          ('discrete_actions', 'discrete_actions_loss_normalizer'), ('continuous_actions', 'continuous_actions_loss_normalizer')):
            if key in agent:
                agent[key] = self._normalize_loss(name, agent[key], upd)
        # the reference's weighted total (dreamer4.py:7708-7723); the per-term weights are the module's buffers (1 or mtp elements)
        wmap = dict(rewards=self.reward_loss_weight, terminals=self.terminal_loss_weight, discrete_actions=self.discrete_action_loss_weight,
                    continuous_actions=self.continuous_action_loss_weight)
        total = flow * self.latent_flow_loss_weight + short * self.shortcut_loss_weight + sum((v * wmap[k].to(v.device)).sum() for k, v in agent.items())
        if not return_all_losses:
            return total
        from collections import namedtuple
        Losses = namedtuple('WorldModelLosses', ('flow', 'shortcut', 'rewards', 'terminals', 'discrete_actions', 'continuous_actions'))
        z = lat.new_zeros(())
        return total, Losses(flow, short, agent.get('rewards', z), agent.get('terminals', z), agent.get('discrete_actions', z), agent.get('continuous_actions', z))

    # ------------------------------------------------------------------------------ generate
    @torch.no_grad()
    def generate(
        self,
        time_steps,
        num_steps=4,
        batch_size=1,
        agent_index=0,
        tasks=None,
        latent_gene_ids=None,
        image_height=None,
        image_width=None,
        return_decoded_video=None,
        context_signal_noise=0.1,
        time_cache: TimeCache | None = None,
        use_time_cache=True,
        return_rewards_per_frame=False,
        return_terminals=False,
        return_agent_actions=False,
        return_log_probs_and_values=False,
        return_for_policy_optimization=False,
        return_time_cache=False,
        store_agent_embed=True,
        store_old_action_unembeds=True,
        prompt=None,
        prompt_latents=None,
        prompt_proprio=None,
        prompt_discrete_actions=None,
        prompt_continuous_actions=None,
        prompt_rewards=None,
        aug_id=False,
        discrete_temperature=1.,
        continuous_temperature=1.,
        *,
        noise: dict | None = None,
        generator: torch.Generator | None = None,
    ):
        """DynamicsWorldModel.generate (dreamer4.py:6308-6774) on the HIP engine.

        Extra keyword-only arguments: `noise` injects the four per-frame random draws
        (`latent`, `context` (F,B,n,dl) normal; `gumbel_u` (F,B,A), `bern_u` (F,B) uniform) for
        parity runs; otherwise they are drawn on the device from `generator`."""
        assert not (prompt is not None and prompt_latents is not None), 'cannot pass in both prompt video and prompt latents'
        if prompt is not None:
            # a video (or image) prompt goes through the tokenizer's encoder                               dreamer4.py:6376-6387
            assert self.video_tokenizer is not None, 'a video prompt needs a video_tokenizer'
            tokenizer = self.video_tokenizer
            if prompt.ndim == 4:
                prompt = prompt.unsqueeze(2)
            if prompt.shape[1] != tokenizer.channels:
                assert prompt.shape[1] == 1
                prompt = prompt.expand(-1, tokenizer.channels, -1, -1, -1)
            prompt_latents = tokenizer.tokenize(prompt)
        return_decoded_video = (self.video_tokenizer is not None) if return_decoded_video is None else return_decoded_video      # dreamer4.py:6695
        if return_decoded_video and self.video_tokenizer is None:
            raise AssertionError('return_decoded_video=True needs a video_tokenizer')
        if latent_gene_ids is not None or prompt_proprio is not None or aug_id not in (False, None, 0):
            raise NotImplementedError('latent genes / proprio / aug conditioning are not implemented')
        assert agent_index == 0
        if return_for_policy_optimization:
            return_agent_actions = return_log_probs_and_values = return_rewards_per_frame = True
            return_terminals = return_terminals or self.predict_terminals
        return_agent_actions = return_agent_actions or return_log_probs_and_values
        assert log2(num_steps).is_integer(), f'number of steps {num_steps} must be a power of 2'
        assert 0 < num_steps <= self.max_steps, f'number of steps {num_steps} must be between 0 and {self.max_steps}'
        if return_agent_actions and not (self.num_discrete_actions or self.num_continuous_actions):
            raise AssertionError('the model has no actions (dreamer4.py:6626)')

        dev, B, T = self.device, batch_size, time_steps
        n, dl = self.latent_shape
        na, A, nc = len(self.num_discrete_actions), sum(self.num_discrete_actions), self.num_continuous_actions
        if isinstance(tasks, int):
            tasks = torch.full((B,), tasks)
        assert tasks is None or tasks.shape[0] == B

        P = 0
        if prompt_latents is not None:
            pl = prompt_latents
            if pl.ndim == 5:
                assert pl.shape[2] == 1
                pl = pl[:, :, 0]
            assert pl.shape[0] == B
            P = pl.shape[1]
        F_ = max(T - P, 0)
        sample_terminals = bool(return_terminals and self.predict_terminals)

        cached = time_cache.frames if time_cache is not None else 0
        if use_time_cache:
            parallel = (P + 1) if (P > 0 and cached == 0) else 1
        else:
            parallel = T
        self._ensure_engine(batch=B, frames=cached + max(T, 1), parallel=parallel)
        lib = _lib.load()
        self._adopt_cache(time_cache if use_time_cache else None, B)

        # ---- noise
        if noise is None:
            g = generator
            noise = dict(
                latent=torch.randn(F_, B, n, dl, device=dev, generator=g),
                context=torch.randn(F_, B, n, dl, device=dev, generator=g) if not use_time_cache else None,
                gumbel_u=torch.rand(F_, B, A, device=dev, generator=g) if return_agent_actions else None,
                bern_u=torch.rand(F_, B, device=dev, generator=g) if sample_terminals else None,
            )
            if return_agent_actions and nc > 0:       # (normal, uniform) per rejection round of the two gammas behind each Beta draw
                noise['beta'] = torch.stack((torch.randn(F_, B, nc, 2, 6, device=dev, generator=g),
                                             torch.rand(F_, B, nc, 2, 6, device=dev, generator=g).clamp(1e-6, 1. - 1e-6)), dim=-1)
        nz = {k: (v.to(dev).float().contiguous() if v is not None else None) for k, v in noise.items()}
        assert nz['latent'].shape[0] >= F_

        # ---- histories
        latents = torch.zeros(B, T, n, dl, device=dev)
        ctx_hist = None
        if P > 0:
            latents[:, :P] = pl.to(dev)
        if not use_time_cache:
            ctx_hist = latents.clone()
        actions = None
        if na > 0 and (return_agent_actions or prompt_discrete_actions is not None):
            actions = torch.zeros(B, T, na, dtype=torch.long, device=dev)
            if prompt_discrete_actions is not None:
                pa = prompt_discrete_actions.to(dev)
                pa = pa[..., None] if pa.ndim == 2 else pa
                actions[:, :pa.shape[1]] = pa[:, :T]
        actions_c = None
        if nc > 0 and (return_agent_actions or prompt_continuous_actions is not None):
            actions_c = torch.zeros(B, T, nc, device=dev)
            if prompt_continuous_actions is not None:
                pc = prompt_continuous_actions.to(dev).float()
                pc = pc[..., None] if pc.ndim == 2 else pc
                actions_c[:, :pc.shape[1]] = pc[:, :T]
        rewards = torch.zeros(B, T, device=dev)
        if prompt_rewards is not None:
            rewards[:, :prompt_rewards.shape[1]] = prompt_rewards.to(dev)[:, :T]
        agent_embed = torch.empty(B, F_, self.dim, device=dev)
        log_probs = torch.empty(B, F_, na, device=dev)
        values = torch.empty(B, F_, device=dev)
        logits = torch.empty(B, F_, A, device=dev)
        log_probs_c = torch.empty(B, F_, nc, device=dev)
        cparams = torch.empty(B, F_, nc, 2, device=dev)
        lens = torch.full((B,), T, dtype=torch.long, device=dev)
        terminals = torch.zeros(B, dtype=torch.uint8, device=dev)
        tk = tasks.to(dev).long().contiguous() if tasks is not None else None

        io = _lib.RolloutIO()
        io.batch, io.time_steps, io.prompt_frames, io.num_steps = B, T, P, num_steps
        io.use_time_cache, io.sample_terminals = int(use_time_cache), int(sample_terminals)
        io.sample_actions = int(return_agent_actions)
        io.context_signal_noise, io.discrete_temperature = context_signal_noise, discrete_temperature
        io.continuous_temperature = continuous_temperature
        P_ = _lib.ptr
        io.beta_noise, io.actions_cont, io.log_probs_cont, io.cont_params = P_(nz.get('beta')), P_(actions_c), P_(log_probs_c), P_(cparams)
        io.noise_latent, io.noise_context = P_(nz['latent']), P_(nz.get('context'))
        io.gumbel_u, io.bern_u, io.tasks = P_(nz.get('gumbel_u')), P_(nz.get('bern_u')), P_(tk)
        io.latents, io.actions, io.rewards, io.ctx_hist = P_(latents), P_(actions), P_(rewards), P_(ctx_hist)
        io.agent_embed, io.log_probs, io.values, io.action_logits = P_(agent_embed), P_(log_probs), P_(values), P_(logits)
        io.lens, io.terminals = P_(lens), P_(terminals)
        if F_ > 0:
            _lib.check(lib.d4_rollout(self._engine, C.byref(io), self._stream()))

        self._cache_serial += 1
        first_new = cached if use_time_cache else 0
        self._mark_slots(first_new, max(first_new + (T if cached == 0 else F_), first_new), self._cache_serial)

        # ---- early exit once every trajectory has terminated (dreamer4.py:6681): the engine always runs all
        # frames (no host sync inside the rollout); frames past the reference's break are dropped here, also from the cache.
        # A call that generates ONE frame (the env-wrapper pattern, dreamer4/env.py:445-483) cannot be shortened — a terminal sampled at the
        # only new frame gives lens = T — so it never waits for the device here: the host goes on preparing the next call while this one runs.
        Tp = T
        terminals_b = terminals.bool()
        if sample_terminals and F_ > 1 and bool(terminals_b.all()):
            Tp = int(lens.max().item())
        Fp = Tp - P
        new_cache = None
        if use_time_cache:
            frames_now = lib.d4_engine_cache_frames(self._engine)
            if Tp < T and F_ > 0:
                frames_now -= T - Tp
                _lib.check(lib.d4_engine_cache_reset(self._engine, frames_now))
            new_cache = TimeCache(self, frames_now, B, self._cache_serial)
            self._live_cache = new_cache
        else:
            self._live_cache = None
        latents = latents[:, :Tp].clamp(-1., 1.)
        video = None
        if return_decoded_video:                                   # dreamer4.py:6699-6711
            video = self.video_tokenizer.decode(latents, height=image_height, width=image_width, generator=generator,
                                                noise=(noise or {}).get('video') if isinstance(noise, dict) else None)

        if not (return_rewards_per_frame or return_agent_actions):
            out = video if return_decoded_video else latents       # dreamer4.py:6723-6729
            return (out, new_cache) if return_time_cache else out

        step_mask = torch.arange(Tp, device=dev) < lens[:, None]
        rewards = rewards[:, :Tp]
        gen = Experience(
            latents=latents,
            video=video,
            proprio=None,
            agent_embed=agent_embed[:, :Fp] if store_agent_embed else None,
            old_action_unembeds=Actions(logits[:, :Fp] if na > 0 else None, cparams[:, :Fp] if nc > 0 else None)
            if (return_agent_actions and store_old_action_unembeds) else None,
            step_size=self.max_steps // num_steps,
            agent_index=agent_index,
            lens=lens,
            is_truncated=~terminals_b,
            terminals=terminals_b,
            is_from_world_model=True,
            episode_return=(rewards * step_mask.float()).sum(dim=-1),
            rewards=rewards if return_rewards_per_frame else None,
            actions=Actions(actions[:, :Tp] if na > 0 else None, actions_c[:, :Tp] if nc > 0 else None) if return_agent_actions else None,
            log_probs=Actions(log_probs[:, :Fp] if na > 0 else None, log_probs_c[:, :Fp] if nc > 0 else None) if return_log_probs_and_values else None,
            values=values[:, :Fp] if return_log_probs_and_values else None,
        )
        return (gen, new_cache) if return_time_cache else gen

    # ------------------------------------------------------------------------------ learn
    def learn_from_experience(
        self,
        experience: Experience,
        policy_optim=None,
        value_optim=None,
        only_learn_policy_value_heads=True,
        objective='ppo',
        use_delight_gating=None,
        delight_temperature=None,
        normalize_advantages=None,
        eps=1e-6,
        *,
        process_group=None,
        stats='global',
    ):
        """DynamicsWorldModel.learn_from_experience (dreamer4.py:5893-6305).  Losses and the
        gradients of both heads come from the HIP learner; the returned tensors are autograd-connected
        to the head parameters, so `loss.backward()` / optimiser usage is unchanged.
        `only_learn_policy_value_heads=False` (dreamer4.py:6045-6075) recomputes the agent embeddings with a forward WITH gradient
        through the differentiable HIP trunk blocks, so both losses also reach every world-model parameter.

        Data parallel: with an initialised process group and `stats='global'` the advantage
        statistics and every masked-mean denominator are all-reduced, so N ranks x B_local equal one
        process at B_global; gradients are all-reduced by the caller (DreamTrainer)."""
        from dreamer4_amd.learner import learn
        return learn(self, experience, policy_optim, value_optim, only_learn_policy_value_heads, objective,
                     use_delight_gating, delight_temperature, normalize_advantages, eps, process_group, stats)


def _debug_buffer(model, name, shape):
    """Test hook: copy an engine-internal activation buffer out (see d4_debug_buffer)."""
    lib = _lib.load()
    p = C.c_void_p()
    _lib.check(lib.d4_debug_buffer(model._engine, name.encode(), C.byref(p)))
    base = model._ws.data_ptr()
    off = p.value - base
    numel = 1
    for s in shape:
        numel *= s
    return model._ws[off:off + 4 * numel].view(torch.float32).view(*shape).clone()
