"""Differentiable trunk blocks on the HIP kernels and the dynamics training forward composed from them.

`feedforward`, `space_attention`, `time_attention` and `cross_attention` are the reference FeedForward (dreamer4/dreamer4.py:2079-2116)
and Attention (dreamer4.py:1968-2075: within a frame, along time with rotary + causal mask, over a context) as
`torch.autograd.Function`s over the C-ABI operators `d4_ff_* / d4_space_attn_* / d4_time_attn_* / d4_cross_attn_*` (include/d4hip.h).
Parameters are passed in the reference's own layout (the tensors of its state_dict), gradients come back in the same layout.  BY DEFAULT the blocks
run as these plain `autograd.Function`s (less host dispatch time; not traceable by torch.compile); with D4_TRUNK_DISPATCHER=1 they go through
their dispatcher registrations `torch.ops.d4hip.{swiglu_ff, attn_block_space, attn_block_time, attn_block_cross}` (dreamer4_amd/ops.py: fake
implementations + registered autograd, same C-ABI operators, same results).  The bare RMSNorm + Linear pieces of the trunk always run on
`torch.ops.d4hip.{rmsnorm, linear}`.  By default a
block keeps the workspace its forward ran in and the backward (`*_backward_saved`) recomputes nothing; with D4_TRUNK_SAVE_FORWARD=0 the
backward recomputes the forward intermediates and nothing but the inputs is kept alive between the two passes.  `transformer` composes
the AxialSpaceTimeTransformer (dreamer4.py:2927-3267), `world_model_prediction` the dynamics model's `get_prediction`
(dreamer4.py:7156-7287), `dynamics_flow_losses` / `dynamics_agent_losses` the losses of the training forward (dreamer4.py:7335-7598).
fp32, HIP device only — there is no CPU fallback.  Not used by the imagination path."""
from __future__ import annotations

import ctypes as C

import torch

from dreamer4_amd import _lib
from dreamer4_amd import ops as _ops  # noqa: F401  (registers torch.ops.d4hip.*)


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _prep(*ts):
    out = []
    for t in ts:
        if t is None:
            out.append(None)
            continue
        if t.device.type != 'cuda':
            raise _lib.D4Error('trunk_ops run only on an MI355X (HIP) device: there is no CPU fallback')
        assert t.dtype == torch.float32, 'fp32 only'
        out.append(t.contiguous())
    return out


def save_forward_workspace():
    """D4_TRUNK_SAVE_FORWARD=0: the backward of every block recomputes its forward (nothing but the inputs is kept between the passes);
    default: each block keeps the workspace its forward ran in (normalised input, concatenated weight images, projections: 0.1-0.2 GB per
    block at 3840 token rows — 288 GB of HBM is what makes that the default) and the backward runs on it without recomputing anything."""
    import os
    return os.environ.get('D4_TRUNK_SAVE_FORWARD', '1') != '0'


def via_dispatcher():
    """D4_TRUNK_DISPATCHER=1: the blocks go through their `torch.ops.d4hip.*` registrations (dispatcher-visible, traceable by
    torch.compile); default: the same C entry points through plain `torch.autograd.Function`s — the dispatcher route costs ~9 % of a
    training step at cfg 2's architecture in host-side dispatch (20.9 vs 19.1 ms, same box), so eager training does not pay for it."""
    import os
    return os.environ.get('D4_TRUNK_DISPATCHER', '0') == '1'


def _workspace(nbytes, device):
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    base = ws.data_ptr()
    return ws, C.c_void_p(base + (-base) % 256)


class _FeedForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, norm_w, w_in, b_in, w_out, b_out):
        x, norm_w, w_in, b_in, w_out, b_out = _prep(x, norm_w, w_in, b_in, w_out, b_out)
        D, inner = x.shape[-1], w_out.shape[1]
        assert w_in.shape == (2 * inner, D) and w_out.shape == (D, inner) and b_in.shape == (2 * inner,) and b_out.shape == (D,)
        rows = x.numel() // D
        lib = _lib.load()
        nbytes = lib.d4_ff_workspace_bytes(rows, D, inner)
        ws, wp = _workspace(nbytes, x.device)
        y = torch.empty_like(x)
        _lib.check(lib.d4_ff_forward(_lib.ptr(x), _lib.ptr(norm_w), _lib.ptr(w_in), _lib.ptr(b_in), _lib.ptr(w_out), _lib.ptr(b_out),
                                     rows, D, inner, _lib.ptr(y), wp, nbytes, _stream(x)))
        ctx.save_for_backward(x, norm_w, w_in, b_in, w_out)
        ctx.fwd_ws = (ws, wp, nbytes) if save_forward_workspace() and any(ctx.needs_input_grad) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, norm_w, w_in, b_in, w_out = ctx.saved_tensors
        (dy,) = _prep(dy)
        D, inner = x.shape[-1], w_out.shape[1]
        rows = x.numel() // D
        lib = _lib.load()
        if ctx.fwd_ws is not None:
            (ws, wp, nbytes), fn = ctx.fwd_ws, lib.d4_ff_backward_saved
        else:
            nbytes = lib.d4_ff_workspace_bytes(rows, D, inner)
            (ws, wp), fn = _workspace(nbytes, x.device), lib.d4_ff_backward
        dx, dn, dwi, dbi, dwo = torch.empty_like(x), torch.empty_like(norm_w), torch.empty_like(w_in), torch.empty_like(b_in), torch.empty_like(w_out)
        dbo = torch.empty(D, device=x.device)
        _lib.check(fn(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(norm_w), _lib.ptr(w_in), _lib.ptr(b_in), _lib.ptr(w_out), rows, D, inner,
                                      _lib.ptr(dx), _lib.ptr(dn), _lib.ptr(dwi), _lib.ptr(dbi), _lib.ptr(dwo), _lib.ptr(dbo), wp, nbytes, _stream(x)))
        return dx, dn, dwi, dbi, dwo, dbo


class _SpaceAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, softclamp, num_special, belief):
        x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma = _prep(x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma)
        assert x.ndim == 3, 'x must be (frames, tokens, dim)'
        F_, S, D = x.shape
        heads, dh = gamma.shape
        assert wq.shape == (heads * dh, D) and wo.shape == (D, heads * dh) and wg.shape == (heads, D)
        assert rv is None or rv.shape == (F_, S, heads, dh)
        lib = _lib.load()
        nbytes = lib.d4_attn_workspace_bytes(F_, S, D, heads, dh)
        ws, wp = _workspace(nbytes, x.device)
        y = torch.empty_like(x)
        _lib.check(lib.d4_space_attn_forward(_lib.ptr(x), _lib.ptr(rv), _lib.ptr(norm_w), _lib.ptr(wq), _lib.ptr(wk), _lib.ptr(wv), _lib.ptr(wo),
                                             _lib.ptr(wg), _lib.ptr(wm), _lib.ptr(bm), _lib.ptr(gamma), F_, S, D, heads, dh,
                                             float(softclamp or 0.), int(num_special), int(bool(belief)), _lib.ptr(y), wp, nbytes, _stream(x)))
        ctx.save_for_backward(x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma)
        ctx.cfg = (float(softclamp or 0.), int(num_special), int(bool(belief)))
        ctx.fwd_ws = (ws, wp, nbytes) if save_forward_workspace() and any(ctx.needs_input_grad) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma = ctx.saved_tensors
        (dy,) = _prep(dy)
        F_, S, D = x.shape
        heads, dh = gamma.shape
        lib = _lib.load()
        if ctx.fwd_ws is not None:
            (ws, wp, nbytes), fn = ctx.fwd_ws, lib.d4_space_attn_backward_saved
        else:
            nbytes = lib.d4_attn_workspace_bytes(F_, S, D, heads, dh)
            (ws, wp), fn = _workspace(nbytes, x.device), lib.d4_space_attn_backward
        e = torch.empty_like
        dx, dn, dq, dk, dv, do, dg, dgam = e(x), e(norm_w), e(wq), e(wk), e(wv), e(wo), e(wg), e(gamma)
        drv, dwm, dbm = (e(rv), e(wm), e(bm)) if rv is not None else (None, None, None)
        _lib.check(fn(
            _lib.ptr(x), _lib.ptr(rv), _lib.ptr(dy), _lib.ptr(norm_w), _lib.ptr(wq), _lib.ptr(wk), _lib.ptr(wv), _lib.ptr(wo), _lib.ptr(wg),
            _lib.ptr(wm), _lib.ptr(bm), _lib.ptr(gamma), F_, S, D, heads, dh, *ctx.cfg,
            _lib.ptr(dx), _lib.ptr(drv), _lib.ptr(dn), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(do), _lib.ptr(dg), _lib.ptr(dwm), _lib.ptr(dbm),
            _lib.ptr(dgam), wp, nbytes, _stream(x)))
        return dx, drv, dn, dq, dk, dv, do, dg, dwm, dbm, dgam, None, None, None


class _TimeAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq, softclamp, belief):
        x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq = _prep(x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq)
        assert x.ndim == 4, 'x must be (batch, frames, tokens, dim)'
        B, T, S, D = x.shape
        heads, dh = gamma.shape
        assert wq.shape == (heads * dh, D) and wo.shape == (D, heads * dh) and wg.shape == (heads, D) and inv_freq.shape == (dh // 2,)
        assert rv is None or rv.shape == (B, T, S, heads, dh)
        lib = _lib.load()
        nbytes = lib.d4_time_attn_workspace_bytes(B, T, S, D, heads, dh)
        ws, wp = _workspace(nbytes, x.device)
        y = torch.empty_like(x)
        _lib.check(lib.d4_time_attn_forward(_lib.ptr(x), _lib.ptr(rv), _lib.ptr(norm_w), _lib.ptr(wq), _lib.ptr(wk), _lib.ptr(wv), _lib.ptr(wo),
                                            _lib.ptr(wg), _lib.ptr(wm), _lib.ptr(bm), _lib.ptr(gamma), _lib.ptr(inv_freq), B, T, S, D, heads, dh,
                                            float(softclamp or 0.), int(bool(belief)), _lib.ptr(y), wp, nbytes, _stream(x)))
        ctx.save_for_backward(x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq)
        ctx.cfg = (float(softclamp or 0.), int(bool(belief)))
        ctx.fwd_ws = (ws, wp, nbytes) if save_forward_workspace() and any(ctx.needs_input_grad) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq = ctx.saved_tensors
        (dy,) = _prep(dy)
        B, T, S, D = x.shape
        heads, dh = gamma.shape
        lib = _lib.load()
        if ctx.fwd_ws is not None:
            (ws, wp, nbytes), fn = ctx.fwd_ws, lib.d4_time_attn_backward_saved
        else:
            nbytes = lib.d4_time_attn_workspace_bytes(B, T, S, D, heads, dh)
            (ws, wp), fn = _workspace(nbytes, x.device), lib.d4_time_attn_backward
        e = torch.empty_like
        dx, dn, dq, dk, dv, do, dg, dgam = e(x), e(norm_w), e(wq), e(wk), e(wv), e(wo), e(wg), e(gamma)
        drv, dwm, dbm = (e(rv), e(wm), e(bm)) if rv is not None else (None, None, None)
        _lib.check(fn(
            _lib.ptr(x), _lib.ptr(rv), _lib.ptr(dy), _lib.ptr(norm_w), _lib.ptr(wq), _lib.ptr(wk), _lib.ptr(wv), _lib.ptr(wo), _lib.ptr(wg),
            _lib.ptr(wm), _lib.ptr(bm), _lib.ptr(gamma), _lib.ptr(inv_freq), B, T, S, D, heads, dh, *ctx.cfg,
            _lib.ptr(dx), _lib.ptr(drv), _lib.ptr(dn), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(do), _lib.ptr(dg), _lib.ptr(dwm), _lib.ptr(dbm),
            _lib.ptr(dgam), wp, nbytes, _stream(x)))
        return dx, drv, dn, dq, dk, dv, do, dg, dwm, dbm, dgam, None, None, None


class _CrossAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx_, q_tokens, context, norm_w, norm_ctx_w, wq, wk, wv, wo, wg, gamma, item_major, softclamp):
        q_tokens, context, norm_w, norm_ctx_w, wq, wk, wv, wo, wg, gamma = _prep(q_tokens, context, norm_w, norm_ctx_w, wq, wk, wv, wo, wg, gamma)
        assert q_tokens.ndim == 3 and context.ndim == 3
        G, nq, D = q_tokens.shape
        nk, Dc = (context.shape[0], context.shape[2]) if item_major else (context.shape[1], context.shape[2])
        assert (context.shape[1] if item_major else context.shape[0]) == G, 'context groups do not match the queries'
        heads, dh = gamma.shape
        assert wq.shape == (heads * dh, D) and wk.shape == (heads * dh, Dc) and wv.shape == (heads * dh, Dc) and wo.shape == (D, heads * dh)
        lib = _lib.load()
        nbytes = lib.d4_cross_attn_workspace_bytes(G, nq, nk, D, Dc, heads, dh)
        ws, wp = _workspace(nbytes, q_tokens.device)
        y = torch.empty_like(q_tokens)
        _lib.check(lib.d4_cross_attn_forward(_lib.ptr(q_tokens), _lib.ptr(context), _lib.ptr(norm_w), _lib.ptr(norm_ctx_w), _lib.ptr(wq), _lib.ptr(wk),
                                             _lib.ptr(wv), _lib.ptr(wo), _lib.ptr(wg), _lib.ptr(gamma), G, nq, nk, int(bool(item_major)), D, Dc, heads, dh,
                                             float(softclamp or 0.), _lib.ptr(y), wp, nbytes, _stream(q_tokens)))
        ctx_.save_for_backward(q_tokens, context, norm_w, norm_ctx_w, wq, wk, wv, wo, wg, gamma)
        ctx_.cfg = (G, nq, nk, int(bool(item_major)), D, Dc, heads, dh, float(softclamp or 0.))
        ctx_.fwd_ws = (ws, wp, nbytes) if save_forward_workspace() and any(ctx_.needs_input_grad) else None
        return y

    @staticmethod
    def backward(ctx_, dy):
        q_tokens, context, norm_w, norm_ctx_w, wq, wk, wv, wo, wg, gamma = ctx_.saved_tensors
        (dy,) = _prep(dy)
        G, nq, nk, item_major, D, Dc, heads, dh, softclamp = ctx_.cfg
        lib = _lib.load()
        if ctx_.fwd_ws is not None:
            (ws, wp, nbytes), fn = ctx_.fwd_ws, lib.d4_cross_attn_backward_saved
        else:
            nbytes = lib.d4_cross_attn_workspace_bytes(G, nq, nk, D, Dc, heads, dh)
            (ws, wp), fn = _workspace(nbytes, q_tokens.device), lib.d4_cross_attn_backward
        e = torch.empty_like
        dq_t, dc, dn, dq, dk, dv, do, dg, dgam = e(q_tokens), e(context), e(norm_w), e(wq), e(wk), e(wv), e(wo), e(wg), e(gamma)
        dnc = e(norm_ctx_w) if norm_ctx_w is not None else None
        _lib.check(fn(
            _lib.ptr(q_tokens), _lib.ptr(context), _lib.ptr(dy), _lib.ptr(norm_w), _lib.ptr(norm_ctx_w), _lib.ptr(wq), _lib.ptr(wk), _lib.ptr(wv),
            _lib.ptr(wo), _lib.ptr(wg), _lib.ptr(gamma), G, nq, nk, item_major, D, Dc, heads, dh, softclamp,
            _lib.ptr(dq_t), _lib.ptr(dc), _lib.ptr(dn), _lib.ptr(dnc), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(do), _lib.ptr(dg), _lib.ptr(dgam),
            wp, nbytes, _stream(q_tokens)))
        return dq_t, dc, dn, dnc, dq, dk, dv, do, dg, dgam, None, None


def _dev(*ts):
    for t in ts:
        if t is not None and t.device.type != 'cuda':
            raise _lib.D4Error('trunk_ops run only on an MI355X (HIP) device: there is no CPU fallback')


def feedforward(x, norm_weight, proj_in_weight, proj_in_bias, proj_out_weight, proj_out_bias):
    """FeedForward.forward (dreamer4.py:2105-2116): proj_out(a * silu(g)), [a | g] = proj_in(RMSNorm(x)).  x (..., dim).
    With D4_TRUNK_DISPATCHER=1: torch.ops.d4hip.swiglu_ff (dreamer4_amd/ops.py: dispatcher-visible, autograd registered); default: the autograd.Function above."""
    _dev(x)
    if not via_dispatcher():
        return _FeedForward.apply(x, norm_weight, proj_in_weight, proj_in_bias, proj_out_weight, proj_out_bias)
    return torch.ops.d4hip.swiglu_ff(x, norm_weight, proj_in_weight, proj_in_bias, proj_out_weight, proj_out_bias)[0]


def space_attention(x, norm_weight, to_q, to_k, to_v, to_out, to_gates, k_gamma, *, residual_values=None, mix_weight=None, mix_bias=None,
                    softclamp_value=50., num_special=1, belief=True):
    """Attention.forward (dreamer4.py:1968-2075), self attention within each frame: x (frames, tokens, dim) -> (frames, tokens, dim).
    `residual_values` (frames, tokens, heads, dim_head) with `mix_weight` / `mix_bias` = to_learned_value_residual_mix.0 (every layer
    but the first); `num_special` trailing tokens are hidden from ordinary queries (dreamer4.py:1769-1783).  (D4_TRUNK_DISPATCHER=1: torch.ops.d4hip.attn_block_space.)"""
    _dev(x)
    assert x.ndim == 3, 'x must be (frames, tokens, dim)'
    if not via_dispatcher():
        return _SpaceAttention.apply(x, residual_values, norm_weight, to_q, to_k, to_v, to_out, to_gates, mix_weight, mix_bias, k_gamma,
                                     softclamp_value, num_special, belief)
    return torch.ops.d4hip.attn_block_space(x, residual_values, norm_weight, to_q, to_k, to_v, to_out, to_gates, mix_weight, mix_bias, k_gamma,
                                            float(softclamp_value or 0.), int(num_special), bool(belief))[0]


def time_attention(x, norm_weight, to_q, to_k, to_v, to_out, to_gates, k_gamma, inv_freq, *, residual_values=None, mix_weight=None,
                   mix_bias=None, softclamp_value=50., belief=True):
    """The trunk's time layers (dreamer4.py:3176-3215): causal attention along time for every token column, rotary positions
    (`inv_freq` = time_rotary.inv_freq), no KV cache (the training form).  x (batch, frames, tokens, dim), frames <= 64;
    `residual_values` (batch, frames, tokens, heads, dim_head).  (D4_TRUNK_DISPATCHER=1: torch.ops.d4hip.attn_block_time.)"""
    _dev(x)
    assert x.ndim == 4, 'x must be (batch, frames, tokens, dim)'
    if not via_dispatcher():
        return _TimeAttention.apply(x, residual_values, norm_weight, to_q, to_k, to_v, to_out, to_gates, mix_weight, mix_bias, k_gamma, inv_freq,
                                    softclamp_value, belief)
    return torch.ops.d4hip.attn_block_time(x, residual_values, norm_weight, to_q, to_k, to_v, to_out, to_gates, mix_weight, mix_bias, k_gamma, inv_freq,
                                           float(softclamp_value or 0.), bool(belief))[0]


def cross_attention(q_tokens, context, norm_weight, norm_context_weight, to_q, to_k, to_v, to_out, to_gates, k_gamma, *,
                    context_item_major=False, softclamp_value=None):
    """Attention.forward with a context (dreamer4.py:1968-2075): q_tokens (groups, nq, dim); context (groups, nk, dim_ctx), or
    (nk, groups, dim_ctx) with `context_item_major` (the stack of layer hiddens of the AttentionPool).  nq, nk <= 64.
    (D4_TRUNK_DISPATCHER=1: torch.ops.d4hip.attn_block_cross.)"""
    _dev(q_tokens)
    assert q_tokens.ndim == 3 and context.ndim == 3
    if not via_dispatcher():
        return _CrossAttention.apply(q_tokens, context, norm_weight, norm_context_weight, to_q, to_k, to_v, to_out, to_gates, k_gamma,
                                     context_item_major, softclamp_value)
    return torch.ops.d4hip.attn_block_cross(q_tokens, context, norm_weight, norm_context_weight, to_q, to_k, to_v, to_out, to_gates, k_gamma,
                                            bool(context_item_major), float(softclamp_value or 0.))[0]


def _norm_linear(x, norm_w, w, bias=None):
    """nn.Sequential(RMSNorm, Linear) of the reference on the HIP operators (torch.ops.d4hip.rmsnorm / linear, autograd registered)."""
    eps = torch.finfo(torch.float32).eps
    return torch.ops.d4hip.linear(torch.ops.d4hip.rmsnorm(x, norm_w, eps), w, bias, None, 0, 0.)


# ------------------------------------------------------------------------------------------------ the trunk, composed
def _attn_w(W, pre):
    return (W[pre + 'norm.weight'], W[pre + 'to_q.weight'], W[pre + 'to_k.weight'], W[pre + 'to_v.weight'], W[pre + 'to_out.weight'],
            W[pre + 'to_gates.0.weight'], W[pre + 'k_heads_rmsnorm.gamma'])


def _ff(W, pre, x):
    return feedforward(x, W[pre + 'norm.weight'], W[pre + 'proj_in.weight'], W[pre + 'proj_in.bias'], W[pre + 'proj_out.weight'], W[pre + 'proj_out.bias'])


def _pool(W, pre, x, hiddens):
    """Residual(AttentionPool) (dreamer4.py:2143-2177 + 1869): one query per token over the stack of layer hiddens.
    hiddens: a list of layer hiddens, or the stack itself (L, rows, D) — `transformer` grows ONE stack by concatenation, so that in the
    backward every pool's context gradient meets the previous pools' as one (L, rows, D) sum instead of L per-hidden sums per pool."""
    shape = x.shape
    ctx = hiddens if torch.is_tensor(hiddens) else torch.stack([h.reshape(-1, shape[-1]) for h in hiddens], dim=0)     # (L, rows, D): item major
    p = pre + 'fn.attn.'
    nw, wq, wk, wv, wo, wg, gam = _attn_w(W, p)
    out = cross_attention(x.reshape(-1, 1, shape[-1]), ctx, nw, W[p + 'norm_context.weight'], wq, wk, wv, wo, wg, gam, context_item_major=True)
    return x + out.reshape(shape)


def transformer(W, tokens, *, is_time, softclamp_value=50., num_special=1, pre='transformer.'):
    """AxialSpaceTimeTransformer.forward (dreamer4.py:2927-3267, defaults: value residual, attention pools, final special cross
    attention; final RMSNorm when the weights hold one) for training: no KV cache, every block a HIP forward + backward operator;
    only the residual adds, reshapes and the two bare RMSNorm + Linear pieces (value residual projection, final norm) are torch ops.
    W: the reference's trunk parameters by state_dict key (with the prefix `pre`); tokens (batch, frames, tokens, dim); is_time: per
    layer.  Differentiable with respect to tokens and every parameter."""
    from torch.nn import functional as F
    b, t, s, d = tokens.shape
    gamma0 = W[pre + 'layers.0.2.fn.k_heads_rmsnorm.gamma']
    h, dh = gamma0.shape
    eps = torch.finfo(torch.float32).eps
    vres = _norm_linear(tokens, W[pre + 'to_value_residual.0.weight'], W[pre + 'to_value_residual.1.weight'])
    vres = vres.reshape(b, t, s, h, dh)
    hiddens = tokens.reshape(1, -1, d)                    # the growing stack (L, rows, D) of layer hiddens
    depth = len(is_time)
    for i, tl in enumerate(is_time):
        ap = f'{pre}layers.{i}.2.fn.'
        nw, wq, wk, wv, wo, wg, gam = _attn_w(W, ap)
        mw, mb = W[ap + 'to_learned_value_residual_mix.0.weight'], W[ap + 'to_learned_value_residual_mix.0.bias']
        if tl:
            out = time_attention(tokens, nw, wq, wk, wv, wo, wg, gam, W[pre + 'time_rotary.inv_freq'], residual_values=vres, mix_weight=mw,
                                 mix_bias=mb, softclamp_value=softclamp_value)
        else:
            out = space_attention(tokens.reshape(b * t, s, d), nw, wq, wk, wv, wo, wg, gam, residual_values=vres.reshape(b * t, s, h, dh),
                                  mix_weight=mw, mix_bias=mb, softclamp_value=softclamp_value, num_special=num_special).reshape(b, t, s, d)
        tokens = tokens + out
        after_attn = tokens
        tokens = tokens + _ff(W, f'{pre}layers.{i}.3.fn.', tokens)
        hiddens = torch.cat((hiddens, after_attn.reshape(1, -1, d), tokens.reshape(1, -1, d)), dim=0)
        if i != depth - 1:
            tokens = _pool(W, f'{pre}attn_pools.{i}.', tokens, hiddens)
    # the special tokens cross-attend the ordinary tokens of their frame, then their own feedforward   dreamer4.py:3227-3238
    non_special, special = tokens[:, :, :-num_special], tokens[:, :, -num_special:]
    cp = pre + 'final_special_cross_attn.fn.'
    nw, wq, wk, wv, wo, wg, gam = _attn_w(W, cp)
    out = cross_attention(special.reshape(b * t, num_special, d), non_special.reshape(b * t, s - num_special, d), nw, W[cp + 'norm_context.weight'],
                          wq, wk, wv, wo, wg, gam)
    special = special + out.reshape(b, t, num_special, d)
    special = special + _ff(W, pre + 'final_special_ff.fn.', special)
    tokens = torch.cat((non_special, special), dim=2)
    tokens = _pool(W, pre + 'final_attn_pool.', tokens, hiddens)
    if pre + 'final_norm.weight' in W:
        tokens = torch.ops.d4hip.rmsnorm(tokens, W[pre + 'final_norm.weight'], eps)
    return tokens


# ------------------------------------------------------------------------------------------------ the dynamics model, training form
def _lq_pool(W, pre, x):
    """LearnedQueriesAttentionPool (dreamer4.py:2179-2210): x (..., n, d_ctx) -> (..., num_queries, dim)."""
    lead = x.shape[:-2]
    ctx = x.reshape(-1, *x.shape[-2:])
    queries = W[pre + 'queries']
    q = queries[None].expand(ctx.shape[0], -1, -1)
    p = pre + 'attn.'
    nw, wq, wk, wv, wo, wg, gam = _attn_w(W, p)
    out = cross_attention(q, ctx, nw, W[p + 'norm_context.weight'], wq, wk, wv, wo, wg, gam)
    return out.reshape(*lead, *out.shape[-2:])


def world_model_prediction(W, noised_latents, signal_levels, step_sizes_log2, *, is_time, num_spatial_tokens, num_register_tokens,
                           num_discrete_actions=(), discrete_actions=None, continuous_actions=None, tasks=None, softclamp_value=50.):
    """DynamicsWorldModel's `get_prediction` (dreamer4.py:7156-7287) for training: noised latents (b, t, n, dl), signal_levels (b, t),
    step_sizes_log2 (b,) -> (latent prediction (b, t, n, dl), agent embedding (b, t, dim)).  Token packing, embeddings and the bare
    RMSNorm + Linear of the latent head are torch ops; the trunk and the learned-query pools are the HIP forward + backward blocks.
    W: the model's parameters by reference state_dict key.  Actions are the ones paired with each frame's NEXT transition
    (`shift_action_tokens=True`): (b, t, na) or (b, t-1, na); frame 0 gets a zero action token."""
    from torch.nn import functional as F
    b, t, n, dl = noised_latents.shape
    d = W['signal_levels_embed.weight'].shape[1] * 2
    eps = torch.finfo(torch.float32).eps
    dev = noised_latents.device
    has_actions = 'action_learned_embed' in W and (len(num_discrete_actions) > 0 or W.get('action_embedder.continuous_action_embed.weight', torch.empty(0)).numel() > 0)
    if num_spatial_tokens == n:
        space = torch.ops.d4hip.linear(noised_latents, W['latents_to_spatial_tokens.weight'], W['latents_to_spatial_tokens.bias'], None, 0, 0.)
    else:
        space = _lq_pool(W, 'latents_to_spatial_tokens.', noised_latents)
    sig = W['signal_levels_embed.weight'][signal_levels]
    stp = W['step_size_embed.weight'][step_sizes_log2][:, None].expand(b, t, -1)
    flow_tok = torch.cat((sig, stp), dim=-1)[:, :, None]
    regs = W['register_tokens'].expand(b, t, -1, -1) if num_register_tokens > 0 else noised_latents.new_zeros(b, t, 0, d)
    agent = W['agent_learned_embed'].expand(b, -1, -1)
    if tasks is not None:
        agent = agent + W['task_embed.weight'][tasks][:, None]
    agent = agent[:, None].expand(b, t, -1, -1)
    parts = [flow_tok, space, regs]
    if has_actions:
        emb = None
        if discrete_actions is not None and discrete_actions.shape[1] > 0:
            offs = _action_offsets(tuple(num_discrete_actions), dev)
            emb = W['action_embedder.discrete_action_embed.weight'][discrete_actions + offs].sum(dim=-2)
        if continuous_actions is not None and continuous_actions.shape[1] > 0:
            c = (W['action_embedder.continuous_action_embed.weight'] * continuous_actions[..., None]).sum(dim=-2)
            emb = c if emb is None else emb + c
        if emb is None:
            act = noised_latents.new_zeros(b, t, d)
        else:
            emb = emb + W['action_learned_embed']
            if emb.shape[1] == t:
                emb = emb[:, :-1]
            act = F.pad(emb, (0, 0, 1, 0), value=0.)
        parts.append(act[:, :, None])
    parts.append(agent)
    tokens = transformer(W, torch.cat(parts, dim=2), is_time=is_time, softclamp_value=softclamp_value)
    space_out, agent_embed = tokens[:, :, 1:1 + num_spatial_tokens], tokens[:, :, -1]
    x = torch.ops.d4hip.rmsnorm(space_out, W['to_latent_pred.0.weight'], eps)
    if num_spatial_tokens != n:
        x = _lq_pool(W, 'to_latent_pred.1.', x)
    return torch.ops.d4hip.linear(x, W['to_latent_pred.2.weight'], None, None, 0, 0.), agent_embed


_OFFSETS = {}


def _action_offsets(sizes, dev):
    """First embedding row of each discrete action type; kept per (sizes, device) so that a step uploads nothing (and can be captured)."""
    key = (sizes, str(dev))
    if key not in _OFFSETS:
        acc, offs = 0, []
        for n in sizes:
            offs.append(acc); acc += n
        _OFFSETS[key] = torch.tensor(offs, device=dev)
    return _OFFSETS[key]


def dynamics_flow_losses(W, latents, noise, signal_levels, step_sizes_log2, shortcut_train, *, max_steps, return_agent_embed=False, lens=None, **model):
    """The flow and shortcut-consistency losses of the dynamics training forward (dreamer4.py:6990-7003, 7335-7431; x-space prediction,
    ramp loss weight, no proprio / variable lengths / loss normalisers: the reference defaults).  `model`: the keyword arguments of
    `world_model_prediction`.  Returns (flow_loss, shortcut_loss[, agent_embed of the main prediction]); backward runs through the HIP blocks."""
    from torch.nn import functional as F
    times = signal_levels.float() / max_steps
    tt = times[:, :, None, None]
    noised = noise.lerp(latents, tt)
    pred, agent_embed = world_model_prediction(W, noised, signal_levels, step_sizes_log2, **model)
    flow_losses = F.mse_loss(pred, latents, reduction='none') * (0.9 * times + 0.1)[:, :, None, None]
    if lens is not None:                                  # variable lengths: frames past a trajectory's length leave the means (dreamer4.py:7418-7426)
        lm = torch.arange(latents.shape[1], device=latents.device)[None, :] < lens[:, None]
        sel = lambda x: x[lm]
    else:
        sel = lambda x: x
    if not shortcut_train:
        fl = sel(flow_losses).mean()
        return (fl, latents.new_zeros(()), agent_embed) if return_agent_embed else (fl, latents.new_zeros(()))
    with torch.no_grad():
        half_log2 = step_sizes_log2 - 1
        half = 2 ** half_log2
        first, _ = world_model_prediction(W, noised, signal_levels, half_log2, **model)
        first_flow = (first - noised) / (1. - tt)
        denoised = noised + first_flow * (half[:, None, None, None] / max_steps)
        sig2 = signal_levels + half[:, None]
        second, _ = world_model_prediction(W, denoised, sig2, half_log2, **model)
        second_flow = (second - denoised) / (1. - (sig2.float() / max_steps)[:, :, None, None])
        target = (first_flow + second_flow) / 2
    shortcut_pred = (pred - noised) / (1. - tt)
    shortcut_losses = F.mse_loss(shortcut_pred, target, reduction='none') * (1. - tt) ** 2
    fl, sh = sel(flow_losses).mean(), sel(shortcut_losses).mean()
    return (fl, sh, agent_embed) if return_agent_embed else (fl, sh)


# ------------------------------------------------------------------------------------------------ agent-token losses of the training forward
def _head_mlp(W, pre, x, n_layers, recipe):
    """The heads' normed MLP (x_mlps_pytorch.create_mlp stand-in): 'pre_rms' or 'post_layer'.  Linear and RMSNorm are the
    differentiable HIP operators (torch.ops.d4hip.linear / rmsnorm: d4_gemm, d4_gemm_tn, d4_rmsnorm_backward) — no vendor BLAS on the path."""
    from torch.nn import functional as F
    eps = torch.finfo(torch.float32).eps
    lin = lambda t, w, b: torch.ops.d4hip.linear(t, w, b, None, 0, 0.)
    for i in range(n_layers):
        last = i == n_layers - 1
        if recipe == 'pre_rms':
            x = lin(torch.ops.d4hip.rmsnorm(x, W[f'{pre}layers.{i}.0.weight'], eps), W[f'{pre}layers.{i}.1.weight'], W[f'{pre}layers.{i}.1.bias'])
            if not last:
                x = F.silu(x)
        elif last:
            x = lin(x, W[f'{pre}layers.{i}.weight'], W[f'{pre}layers.{i}.bias'])
        else:
            x = lin(x, W[f'{pre}layers.{i}.0.weight'], W[f'{pre}layers.{i}.0.bias'])
            x = F.silu(F.layer_norm(x, x.shape[-1:], W[f'{pre}layers.{i}.1.weight'], W[f'{pre}layers.{i}.1.bias'], 1e-5))
    return x


def _mtp_targets(t, steps):
    """create_multi_token_prediction_targets (dreamer4.py:530-552)."""
    n = t.shape[1]
    idx = torch.arange(n, device=t.device)[:, None] + torch.arange(steps, device=t.device)[None, :]
    mask = idx < n
    idx = idx.masked_fill(~mask, 0)
    return t[:, idx], mask[None].expand(t.shape[0], -1, -1)


def hl_gauss_probs(values, vrange, num_bins, sigma_to_bin_ratio=2., eps=1e-10):
    """HL-Gauss soft targets (Farebrother et al. 2024; hl_gauss_pytorch stand-in)."""
    support = torch.linspace(vrange[0], vrange[1], num_bins + 1, device=values.device)
    sigma = sigma_to_bin_ratio * (vrange[1] - vrange[0]) / num_bins
    cdf = torch.special.erf((support - values.clamp(vrange[0], vrange[1])[..., None]) / (2. ** 0.5 * sigma))
    z = cdf[..., -1] - cdf[..., 0]
    return (cdf[..., 1:] - cdf[..., :-1]) / z.clamp(min=eps)[..., None]


def dynamics_agent_losses(W, agent_embed, latents, *, multi_token_pred_len, num_discrete_actions, reward_range, reward_num_bins,
                          policy_head_mlp_depth, terminal_mlp_depth, head_mlp_recipe='pre_rms', continuous_beta_param='softplus_p1', gae_discount_factor=0.997,
                          hl_sigma_ratio=2., hl_eps=1e-10, rewards=None, discrete_actions=None, terminals=None, lens=None,
                          continuous_actions=None):
    """The agent-token losses of the training forward (dreamer4.py:7432-7598): multi-token-prediction reward cross entropy against HL-Gauss
    soft targets, terminal BCE with label smoothing, behaviour-cloning log-likelihood of the discrete actions (multi-token prediction,
    `shift_action_tokens=True`).  Every Linear / RMSNorm runs on the differentiable HIP operators (torch.ops.d4hip.linear / rmsnorm: no vendor
    BLAS); the loss algebra on their outputs (log-softmax, masks, means) is elementwise torch on the device; gradients reach the
    trunk through `agent_embed`.  Returns a dict of the terms that were asked for: rewards (mtp,), terminals (), discrete_actions (mtp,), continuous_actions (mtp,)."""
    from torch.nn import functional as F
    out = {}
    mtp = multi_token_pred_len
    eps = torch.finfo(torch.float32).eps
    t = agent_embed.shape[1]
    lm = (torch.arange(t, device=agent_embed.device)[None, :] < lens[:, None]) if lens is not None else None     # dreamer4.py:7418-7423
    if rewards is not None:
        two_hot = hl_gauss_probs(rewards, reward_range, reward_num_bins, hl_sigma_ratio, hl_eps)
        x = agent_embed[:, :-1]
        # the ensemble of mtp [RMSNorm -> Linear] members as ONE HIP GEMM: the members' norm gains are folded into their weights (as the engine
        # does at prepare), so a single unit-gain normalisation of the rows feeds a [mtp * bins][D] weight            dreamer4.py:7440-7450
        g_, w_ = W['to_reward_pred.params.0'], W['to_reward_pred.params.1']                                    # [mtp][D], [mtp][bins][D]
        xhat = torch.ops.d4hip.rmsnorm(x.contiguous(), torch.ones_like(g_[0]), eps)
        pred = torch.ops.d4hip.linear(xhat, (w_ * g_[:, None, :]).reshape(-1, w_.shape[-1]), None, None, 0, 0.).unflatten(-1, (mtp, w_.shape[1]))
        tgt, mask = _mtp_targets(two_hot[:, 1:], mtp)
        losses = -(tgt * pred.log_softmax(dim=-1)).sum(dim=-1).masked_fill(~mask, 0.)
        out['rewards'] = losses[lm[:, :-1]].mean(dim=0) if lm is not None else losses.mean(dim=(0, 1))
    if terminals is not None:
        pooled = latents[:, 1:].mean(dim=-2)
        logit = _head_mlp(W, 'to_state_terminal_pred.0.', pooled, terminal_mlp_depth + 2, head_mlp_recipe).squeeze(-1)
        e = 1. - gae_discount_factor
        tl = F.binary_cross_entropy_with_logits(logit, terminals[:, 1:].float().clamp(min=e, max=1. - e), reduction='none')
        out['terminals'] = tl[lm[:, :-1]].mean() if lm is not None else tl.mean()
    if (discrete_actions is not None or continuous_actions is not None) and t > 1:
        pe = _head_mlp(W, 'policy_head.', agent_embed, policy_head_mlp_depth + 2, head_mlp_recipe)
    if continuous_actions is not None and t > 1:
        # Beta log-likelihood (dreamer4.py:7566-7597); alpha = link(raw0) + 1, beta = link(raw1) + 1 with the link named by
        # `continuous_beta_param` (world_model.BETA_PARAMS: the stand-in's softplus, or exp)
        padded = F.pad(continuous_actions, (0, 0, 1, 0), value=0.)
        tgt, mask = _mtp_targets(padded, mtp)
        tgt, mask = tgt[:, 1:].clamp(1e-5, 1. - 1e-5), mask[:, 1:]
        wc = W['action_embedder.continuous_action_unembed']                                                    # [nc][mtp][4 D][2]
        nc_ = wc.shape[0]
        # every member's (alpha, beta) rows in one HIP GEMM: weight [mtp][nc][2][4 D], member-major
        params = torch.ops.d4hip.linear(pe, wc.permute(1, 0, 3, 2).reshape(-1, wc.shape[2]), None, None, 0, 0.).unflatten(-1, (mtp, nc_, 2))       # (b, t, mtp, nc, 2)
        link = torch.exp if continuous_beta_param == 'exp_p1' else F.softplus
        a, b_ = link(params[..., 0]) + 1., link(params[..., 1]) + 1.
        lp = (a - 1.) * torch.log(tgt) + (b_ - 1.) * torch.log1p(-tgt) + torch.lgamma(a + b_) - torch.lgamma(a) - torch.lgamma(b_)
        nl = (-lp).masked_fill(~mask[..., None], 0.)                                                            # (b, t, mtp, nc)
        out['continuous_actions'] = nl[lm].mean(dim=(0, 2)) if lm is not None else nl.mean(dim=(0, 1, 3))
    if discrete_actions is not None and t > 1:
        padded = F.pad(discrete_actions, (0, 0, 1, 0), value=-1)
        tgt, mask = _mtp_targets(padded, mtp)
        tgt, mask = tgt[:, 1:].clamp(min=0), mask[:, 1:]
        # all mtp members in ONE HIP GEMM ([mtp * A][4 D] weight, member-major) and one pass of the loss algebra over (b, t, member): the per-member
        # python loop was ~10 tiny launches per member and action type                                              dreamer4.py:7520-7560
        un = W['action_embedder.discrete_action_unembed']                                                           # [A][mtp][4 D]
        logits = torch.ops.d4hip.linear(pe, un.transpose(0, 1).reshape(-1, un.shape[-1]), None, None, 0, 0.).unflatten(-1, (mtp, un.shape[0]))   # (b, t, mtp, A)
        lps, o = [], 0
        for a, n in enumerate(num_discrete_actions):
            lp = logits[..., o:o + n].log_softmax(dim=-1)
            lps.append(lp.gather(-1, tgt[..., a:a + 1]).squeeze(-1))
            o += n
        nl = (-torch.stack(lps, dim=-1)).masked_fill(~mask[..., None], 0.)                                          # (b, t, mtp, na)
        out['discrete_actions'] = nl[lm].mean(dim=(0, 2)) if lm is not None else nl.mean(dim=(0, 1, 3))
    return out
