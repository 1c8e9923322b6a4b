"""Host-side mirror of the reference `VideoTokenizer` for its two inference paths:
`decode(latents)` — what `DynamicsWorldModel.generate(return_decoded_video=True)` calls (dreamer4/dreamer4.py:4186-4237, 6694-6711) —
and `tokenize(video)` — what `generate(prompt=video)` and `DynamicsWorldModel.forward(video=...)` call (dreamer4.py:4107-4113,
6376-6387).

Same constructor keyword names as the reference (dreamer4.py:3686-3764), the same state_dict key names for the parameters
(`latents_to_decoder.*`, `time_embed.*`, `noised_patch_to_tokens.*`, `decoder.*`; `patch_to_tokens.*`, `latent_tokens`,
`encoder_transformer.*`, `encoded_to_latents.*`), so a reference tokenizer checkpoint loads with `load_state_dict`.  The compute is
the HIP engine in decoder / encoder mode (include/d4hip.h `d4_decoder_forward`, `d4_encoder_forward`): the dynamics model's trunk
kernels over [patches | latent tokens] per frame, a wide attention kernel for the ~100-token space layers, patchify / un-patchify
kernels.  The training forward (reconstruction / LPIPS / time-decorrelation losses) is out of scope and raises.  Supported subset =
the reference defaults: flow decoder with `decoder_flow_steps` Euler steps, no slot attention, causal conv, MOSS, aug conditioning,
PoPE, separate flow decoder, BYOL."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from dreamer4_amd import _lib
from dreamer4_amd.checkpoint import SaveLoad
from dreamer4_amd.world_model import MLP_RECIPES, _linear_b, _linear_w, _register, mlp_param_specs, mlp_widths

_UNSUPPORTED = dict(
    latent_init_patch_size=None, encoder_full_spatial_attn=False, decoder_full_spatial_attn=False, attn_kwargs={}, ff_kwargs={},
    use_causal_conv3d=False, use_shifted_patch_tokenization=False, encoder_moss_layers=(), decoder_moss_layers=(), use_time_rnn=False,
    time_attention_use_pope=False, space_attention_use_pope=False, h_net_layer=None, slot_attention_initted_latents=False,
    decoder_slot_attention_initted_spatial_tokens=False, mot_temporal=False, has_aug_conditioning=False, separate_flow_decoder=False,
    encode_temporal_diff=False, has_byol=False, decoder_pos_emb_mlp_activation='silu', attn_softclamp_value=50.,
)


class VideoTokenizer(SaveLoad, nn.Module):
    def __init__(self, dim, dim_latent, patch_size, image_size=None, image_height=None, image_width=None, num_latent_tokens=64,
                 encoder_depth=4, decoder_depth=4, time_block_every=4, attn_dim_head=64, attn_heads=8, decoder_pos_mlp_depth=2,
                 channels=3, decoder_flow_steps=1, head_mlp_recipe='pre_rms', **kwargs):
        config = dict(locals())
        super().__init__()
        self._record_config(config)
        for k, v in kwargs.items():
            if k in _UNSUPPORTED:
                if v != _UNSUPPORTED[k]:
                    raise NotImplementedError(f'{k}={v!r} is outside the supported decode subset (reference defaults only)')
            elif k not in ('lpips_loss_network', 'lpips_loss_weight', 'per_image_patch_mask_prob', 'use_loss_normalization'):
                raise NotImplementedError(f'{k!r}: the tokenizer mirror implements the decode path only')
        assert image_size is not None or (image_height is not None and image_width is not None), \
            'either image_size or both image_height and image_width must be provided'
        if decoder_flow_steps < 1:
            raise NotImplementedError('decoder_flow_steps = 0 (no flow decoder) is not implemented')
        if attn_dim_head != 64:
            raise NotImplementedError('the decoder engine needs attn_dim_head == 64')
        self.dim, self.dim_latent, self.patch_size, self.channels = dim, dim_latent, patch_size, channels
        self.image_height = image_height if image_height is not None else image_size
        self.image_width = image_width if image_width is not None else image_size
        self.num_latent_tokens, self.decoder_depth, self.time_block_every = num_latent_tokens, decoder_depth, time_block_every
        self.encoder_depth = encoder_depth
        self.attn_heads, self.attn_dim_head = attn_heads, attn_dim_head
        self.decoder_pos_mlp_depth, self.decoder_flow_steps = decoder_pos_mlp_depth, decoder_flow_steps
        self.head_mlp_recipe = head_mlp_recipe
        self.latent_shape = (num_latent_tokens, dim_latent)
        self.ff_inner = int(dim * 4 * 2 / 3)
        self._build_parameters()
        self._engines = {}          # mode -> dict(engine, caps, ws, bound_sig, prep_version)

    def _build_parameters(self):
        D, dl, h, dh = self.dim, self.dim_latent, self.attn_heads, self.attn_dim_head
        hd, dp = h * dh, self.channels * self.patch_size ** 2
        reg = lambda k, t: _register(self, k, t)

        def attn(pre, heads, ctx_norm, mix, dh_=dh):
            inner = heads * dh_
            reg(pre + 'norm.weight', torch.ones(D))
            if ctx_norm:
                reg(pre + 'norm_context.weight', torch.ones(D))
            for nm, shape in (('to_q', (inner, D)), ('to_k', (inner, D)), ('to_v', (inner, D)), ('to_out', (D, inner))):
                reg(pre + nm + '.weight', _linear_w(*shape))
            reg(pre + 'to_gates.0.weight', _linear_w(heads, D))
            reg(pre + 'k_heads_rmsnorm.gamma', torch.zeros(heads, dh_))
            if mix:
                reg(pre + 'to_learned_value_residual_mix.0.weight', _linear_w(heads, D))
                reg(pre + 'to_learned_value_residual_mix.0.bias', _linear_b(heads, D))

        def ff(pre):
            reg(pre + 'norm.weight', torch.ones(D))
            reg(pre + 'proj_in.weight', _linear_w(2 * self.ff_inner, D)); reg(pre + 'proj_in.bias', _linear_b(2 * self.ff_inner, D))
            reg(pre + 'proj_out.weight', _linear_w(D, self.ff_inner)); reg(pre + 'proj_out.bias', _linear_b(D, self.ff_inner))

        reg('latents_to_decoder.weight', _linear_w(D, dl))
        reg('time_embed.weight', torch.randn(self.decoder_flow_steps, D))
        reg('noised_patch_to_tokens.1.weight', _linear_w(D, dp)); reg('noised_patch_to_tokens.1.bias', _linear_b(D, dp))
        reg('noised_patch_to_tokens.2.weight', torch.ones(D))
        fan_in = 2
        for key, shape, kind in mlp_param_specs(self.head_mlp_recipe, mlp_widths(2, 2 * D, D, self.decoder_pos_mlp_depth)):
            if kind == 'norm_w':
                reg('decoder.to_decoder_pos_emb.' + key, torch.ones(shape))
            elif kind == 'norm_b':
                reg('decoder.to_decoder_pos_emb.' + key, torch.zeros(shape))
            elif kind == 'lin_w':
                reg('decoder.to_decoder_pos_emb.' + key, _linear_w(*shape)); fan_in = shape[1]
            else:
                reg('decoder.to_decoder_pos_emb.' + key, _linear_b(shape[0], fan_in))
        reg('decoder.tokens_to_patch.0.weight', _linear_w(dp, D)); reg('decoder.tokens_to_patch.0.bias', _linear_b(dp, D))

        def trunk(tp, depth):
            inv_freq = 1.0 / (10000. ** (torch.arange(0, dh, 2).float() / dh))
            _register(self, tp + 'time_rotary.inv_freq', inv_freq, buffer=True)
            reg(tp + 'to_value_residual.0.weight', torch.ones(D)); reg(tp + 'to_value_residual.1.weight', _linear_w(hd, D))
            for i in range(depth):
                attn(f'{tp}layers.{i}.2.fn.', h, False, True)
                ff(f'{tp}layers.{i}.3.fn.')
            for i in range(depth - 1):
                attn(f'{tp}attn_pools.{i}.fn.attn.', 4, True, False, 64)
            attn(tp + 'final_attn_pool.fn.attn.', 4, True, False, 64)
            reg(tp + 'final_norm.weight', torch.ones(D))
            attn(tp + 'final_special_cross_attn.fn.', h, True, True)
            ff(tp + 'final_special_ff.fn.')

        # the decoder trunk's final special cross-attention / feedforward (dreamer4.py:2901-2907) update only the last latent token,
        # which the decoder never reads back: the parameters exist for key parity and are not bound to the decoder engine
        trunk('decoder.transformer.', self.decoder_depth)
        # encoder (dreamer4.py:3796, 3838-3848, 3912-3936)
        reg('latent_tokens', torch.randn(self.num_latent_tokens, D) * 1e-2)
        reg('mask_token', torch.randn(D) * 1e-2)             # training-only (MAE masking, dreamer4.py:4318-4340); key parity
        reg('patch_to_tokens.1.weight', _linear_w(D, dp)); reg('patch_to_tokens.1.bias', _linear_b(D, dp))
        reg('patch_to_tokens.2.weight', torch.ones(D))
        trunk('encoder_transformer.', self.encoder_depth)
        reg('encoded_to_latents.weight', _linear_w(dl, D))
        _register(self, 'zero', torch.tensor(0.), buffer=True, persistent=False)

    @property
    def device(self):
        return self.zero.device

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    _ENCODER_KEYS = ('latent_tokens', 'patch_to_tokens.', 'encoder_transformer.', 'encoded_to_latents.')

    def _engine_tensors(self, mode):
        out = {}
        for k, v in list(self.named_parameters()) + list(self.named_buffers()):
            if k in ('zero', 'mask_token'):
                continue
            enc = k.startswith(self._ENCODER_KEYS)
            if mode == 2 and enc:
                out[k] = v
            elif mode == 1 and not enc and 'final_special' not in k:
                out[k] = v
        return out

    def __deepcopy__(self, memo):
        # engine handles are per-instance device state: a copy starts without any (DynamicsWorldModel deep-copies its tokenizer, dreamer4.py:4788)
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k == '_engines' else copy.deepcopy(v, memo)
        return new

    def __del__(self):
        try:
            lib = _lib.load()
            for st in self._engines.values():
                if st['engine'] is not None:
                    lib.d4_engine_destroy(st['engine'])
            self._engines.clear()
        except Exception:
            pass

    def _ensure_engine(self, batch, frames, mode=1):
        if self.device.type != 'cuda':
            raise _lib.D4Error('VideoTokenizer runs only on an MI355X (HIP) device: there is no CPU fallback')
        lib = _lib.load()
        st = self._engines.setdefault(mode, dict(engine=None, caps=None, ws=None, bound_sig=None, prep_version=None))
        if st['caps'] is None or batch > st['caps'][0] or frames > st['caps'][1]:
            old = st['caps'] or (0, 0)
            caps = (max(batch, old[0]), max(frames, old[1]))
            if st['engine'] is not None:
                lib.d4_engine_destroy(st['engine'])
                st['engine'] = None
            c = _lib.Config()
            c.mode = mode
            c.dim, c.dim_latent, c.num_latent_tokens = self.dim, self.dim_latent, self.num_latent_tokens
            c.depth = self.decoder_depth if mode == 1 else self.encoder_depth
            c.time_block_every, c.attn_heads, c.attn_dim_head, c.attn_softclamp_value = self.time_block_every, self.attn_heads, self.attn_dim_head, 50.
            c.num_spatial_tokens, c.num_register_tokens, c.max_steps, c.multi_token_pred_len = 0, 0, 64, 1
            c.reward_num_bins = c.value_num_bins = 3
            c.pool_heads, c.pool_dim_head = 4, 64
            c.head_mlp_recipe = MLP_RECIPES[self.head_mlp_recipe]
            c.patch_size, c.channels, c.image_height, c.image_width = self.patch_size, self.channels, self.image_height, self.image_width
            c.decoder_flow_steps, c.decoder_pos_mlp_depth = self.decoder_flow_steps, self.decoder_pos_mlp_depth
            c.max_batch, c.max_frames, c.max_parallel_frames, c.max_learn_rows = caps[0], caps[1], caps[1], 0
            eng = C.c_void_p()
            _lib.check(lib.d4_engine_create(C.byref(c), C.byref(eng)))
            st['engine'], st['caps'] = eng, caps
            nbytes = lib.d4_engine_workspace_bytes(eng)
            st['ws'] = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            base = st['ws'].data_ptr()
            _lib.check(lib.d4_engine_set_workspace(eng, C.c_void_p(base + (-base) % 256), nbytes))
            st['bound_sig'] = None
        tensors = self._engine_tensors(mode)
        sig = tuple((k, t.data_ptr()) for k, t in tensors.items())
        ver = tuple(t._version for t in tensors.values())
        if sig != st['bound_sig']:
            for k, t in tensors.items():
                assert t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device, k
                _lib.check(lib.d4_engine_bind(st['engine'], k.encode(), _lib.ptr(t), None, t.numel()))
            st['bound_sig'], st['prep_version'] = sig, None
        if ver != st['prep_version']:
            _lib.check(lib.d4_engine_prepare(st['engine'], self._stream()))
            st['prep_version'] = ver
        return st['engine']

    def invalidate_prepared(self):
        for st in self._engines.values():
            st['prep_version'] = None

    # ------------------------------------------------------------------------------------------ decode
    @torch.no_grad()
    def decode(self, latents, height=None, width=None, aug_id=None, return_recons_across_steps=False, *, noise=None, generator=None,
               max_batch=None):
        """VideoTokenizer.decode (dreamer4.py:4186-4237): latents (b, t, n, d) -> video (b, c, t, h, w).  `noise` injects the one
        random draw (the initial flow sample, dreamer4.py:4212) for parity runs; `max_batch` bounds the trajectories decoded per
        engine pass (workspace: ~0.1 MB per token row)."""
        if aug_id not in (None, 0, False):
            raise NotImplementedError('aug conditioning is not implemented')
        if (height not in (None, self.image_height)) or (width not in (None, self.image_width)):
            raise NotImplementedError('decoding at a resolution other than the one given to the constructor is not implemented')
        B, T = latents.shape[:2]
        dev = self.device
        lat = latents.to(dev).float().reshape(B, T, *self.latent_shape).contiguous()
        H, W, Cc = self.image_height, self.image_width, self.channels
        if noise is None:
            noise = torch.randn(B, Cc, T, H, W, device=dev, generator=generator)
        video = noise.to(dev).float().contiguous().clone()
        assert video.shape == (B, Cc, T, H, W)
        chunk = max_batch or B
        lib = _lib.load()
        steps = self.decoder_flow_steps
        preds = []
        for i in range(steps):
            pred = torch.empty_like(video)
            for b0 in range(0, B, chunk):
                b1 = min(B, b0 + chunk)
                eng = self._ensure_engine(b1 - b0, T)
                _lib.check(lib.d4_decoder_forward(eng, _lib.ptr(lat[b0:b1]), _lib.ptr(video[b0:b1]), i, b1 - b0, T, _lib.ptr(pred[b0:b1]), self._stream()))
            # video += (pred - video) / (1 - i / steps) * (1 / steps)          dreamer4.py:4226-4230
            _lib.check(lib.d4_euler_step(_lib.ptr(video), _lib.ptr(pred), video.numel(), 1. - i / steps, 1. / steps, self._stream()))
            if return_recons_across_steps:
                preds.append(pred)
        return (video, preds) if return_recons_across_steps else video

    # ------------------------------------------------------------------------------------------ tokenize
    @torch.no_grad()
    def tokenize(self, video, aug_id=None, *, max_batch=None):
        """VideoTokenizer.tokenize (dreamer4.py:4107-4113): video (b, c, t, h, w) or images (b, c, h, w) -> latents (b, t, n, d) in
        (-1, 1): eval-mode `forward(return_latents=True)` (dreamer4.py:4239-4433, no MAE masking).  `max_batch` bounds the clips
        encoded per engine pass."""
        if aug_id not in (None, 0, False):
            raise NotImplementedError('aug conditioning is not implemented')
        dev = self.device
        if video.ndim == 4:
            video = video.unsqueeze(2)                     # images: a one-frame video (dreamer4.py:4257-4258)
        assert video.ndim == 5, 'video must be (batch, channels, time, height, width)'
        B, Cc, T, H, W = video.shape
        assert Cc == self.channels, f'expected {self.channels} channels, got {Cc}'
        if (H, W) != (self.image_height, self.image_width):
            raise NotImplementedError('tokenizing at a resolution other than the one given to the constructor is not implemented')
        video = video.to(dev).float().contiguous()
        out = torch.empty(B, T, *self.latent_shape, device=dev)
        chunk = max_batch or B
        lib = _lib.load()
        for b0 in range(0, B, chunk):
            b1 = min(B, b0 + chunk)
            eng = self._ensure_engine(b1 - b0, T, mode=2)
            _lib.check(lib.d4_encoder_forward(eng, _lib.ptr(video[b0:b1]), b1 - b0, T, _lib.ptr(out[b0:b1]), self._stream()))
        return out

    def forward(self, *args, **kwargs):
        raise NotImplementedError('the tokenizer mirror implements tokenize / decode; its training forward (reconstruction, LPIPS and '
                                  'time-decorrelation losses, dreamer4.py:4239-4560) is out of scope')
