"""Differentiable trunk blocks on the HIP kernels — the first slice of the trunk backward (SURVEY.md 8f-3 groundwork).

`feedforward` is the reference FeedForward (dreamer4/dreamer4.py:2079-2116) and `space_attention` the reference Attention in its
within-frame self-attention form (dreamer4.py:1968-2075) as `torch.autograd.Function`s over the C-ABI operators
`d4_ff_forward / d4_ff_backward` and `d4_space_attn_forward / d4_space_attn_backward` (include/d4hip.h).  Parameters are passed in the
reference's own layout (the tensors of its state_dict), gradients come back in the same layout, and the backward recomputes the
forward intermediates, so nothing but the inputs is kept alive between the two passes.  fp32, HIP device only — there is no CPU
fallback.  Not used by the imagination path; the dynamics training branch (dreamer4.py:7297-7431) is the consumer to come."""
from __future__ import annotations

import ctypes as C

import torch

from dreamer4_amd import _lib


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
        return y

    @staticmethod
    def backward(ctx, dy):
        x, norm_w, w_in, b_in, w_out = ctx.saved_tensors
        (dy,) = _prep(dy)
        D, inner = x.shape[-1], w_out.shape[1]
        rows = x.numel() // D
        lib = _lib.load()
        nbytes = lib.d4_ff_workspace_bytes(rows, D, inner)
        ws, wp = _workspace(nbytes, x.device)
        dx, dn, dwi, dbi, dwo = torch.empty_like(x), torch.empty_like(norm_w), torch.empty_like(w_in), torch.empty_like(b_in), torch.empty_like(w_out)
        dbo = torch.empty(D, device=x.device)
        _lib.check(lib.d4_ff_backward(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(norm_w), _lib.ptr(w_in), _lib.ptr(b_in), _lib.ptr(w_out), rows, D, inner,
                                      _lib.ptr(dx), _lib.ptr(dn), _lib.ptr(dwi), _lib.ptr(dbi), _lib.ptr(dwo), _lib.ptr(dbo), wp, nbytes, _stream(x)))
        return dx, dn, dwi, dbi, dwo, dbo


def feedforward(x, norm_weight, proj_in_weight, proj_in_bias, proj_out_weight, proj_out_bias):
    """FeedForward.forward (dreamer4.py:2105-2116): proj_out(a * silu(g)), [a | g] = proj_in(RMSNorm(x)).  x (..., dim)."""
    return _FeedForward.apply(x, norm_weight, proj_in_weight, proj_in_bias, proj_out_weight, proj_out_bias)


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
        return y

    @staticmethod
    def backward(ctx, dy):
        x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma = ctx.saved_tensors
        (dy,) = _prep(dy)
        F_, S, D = x.shape
        heads, dh = gamma.shape
        lib = _lib.load()
        nbytes = lib.d4_attn_workspace_bytes(F_, S, D, heads, dh)
        ws, wp = _workspace(nbytes, x.device)
        e = torch.empty_like
        dx, dn, dq, dk, dv, do, dg, dgam = e(x), e(norm_w), e(wq), e(wk), e(wv), e(wo), e(wg), e(gamma)
        drv, dwm, dbm = (e(rv), e(wm), e(bm)) if rv is not None else (None, None, None)
        _lib.check(lib.d4_space_attn_backward(
            _lib.ptr(x), _lib.ptr(rv), _lib.ptr(dy), _lib.ptr(norm_w), _lib.ptr(wq), _lib.ptr(wk), _lib.ptr(wv), _lib.ptr(wo), _lib.ptr(wg),
            _lib.ptr(wm), _lib.ptr(bm), _lib.ptr(gamma), F_, S, D, heads, dh, *ctx.cfg,
            _lib.ptr(dx), _lib.ptr(drv), _lib.ptr(dn), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(do), _lib.ptr(dg), _lib.ptr(dwm), _lib.ptr(dbm),
            _lib.ptr(dgam), wp, nbytes, _stream(x)))
        return dx, drv, dn, dq, dk, dv, do, dg, dwm, dbm, dgam, None, None, None


def space_attention(x, norm_weight, to_q, to_k, to_v, to_out, to_gates, k_gamma, *, residual_values=None, mix_weight=None, mix_bias=None,
                    softclamp_value=50., num_special=1, belief=True):
    """Attention.forward (dreamer4.py:1968-2075), self attention within each frame: x (frames, tokens, dim) -> (frames, tokens, dim).
    `residual_values` (frames, tokens, heads, dim_head) with `mix_weight` / `mix_bias` = to_learned_value_residual_mix.0 (every layer
    but the first); `num_special` trailing tokens are hidden from ordinary queries (dreamer4.py:1769-1783)."""
    return _SpaceAttention.apply(x, residual_values, norm_weight, to_q, to_k, to_v, to_out, to_gates, mix_weight, mix_bias, k_gamma,
                                 softclamp_value, num_special, belief)


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
        return y

    @staticmethod
    def backward(ctx, dy):
        x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq = ctx.saved_tensors
        (dy,) = _prep(dy)
        B, T, S, D = x.shape
        heads, dh = gamma.shape
        lib = _lib.load()
        nbytes = lib.d4_time_attn_workspace_bytes(B, T, S, D, heads, dh)
        ws, wp = _workspace(nbytes, x.device)
        e = torch.empty_like
        dx, dn, dq, dk, dv, do, dg, dgam = e(x), e(norm_w), e(wq), e(wk), e(wv), e(wo), e(wg), e(gamma)
        drv, dwm, dbm = (e(rv), e(wm), e(bm)) if rv is not None else (None, None, None)
        _lib.check(lib.d4_time_attn_backward(
            _lib.ptr(x), _lib.ptr(rv), _lib.ptr(dy), _lib.ptr(norm_w), _lib.ptr(wq), _lib.ptr(wk), _lib.ptr(wv), _lib.ptr(wo), _lib.ptr(wg),
            _lib.ptr(wm), _lib.ptr(bm), _lib.ptr(gamma), _lib.ptr(inv_freq), B, T, S, D, heads, dh, *ctx.cfg,
            _lib.ptr(dx), _lib.ptr(drv), _lib.ptr(dn), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(do), _lib.ptr(dg), _lib.ptr(dwm), _lib.ptr(dbm),
            _lib.ptr(dgam), wp, nbytes, _stream(x)))
        return dx, drv, dn, dq, dk, dv, do, dg, dwm, dbm, dgam, None, None, None


def time_attention(x, norm_weight, to_q, to_k, to_v, to_out, to_gates, k_gamma, inv_freq, *, residual_values=None, mix_weight=None,
                   mix_bias=None, softclamp_value=50., belief=True):
    """The trunk's time layers (dreamer4.py:3176-3215): causal attention along time for every token column, rotary positions
    (`inv_freq` = time_rotary.inv_freq), no KV cache (the training form).  x (batch, frames, tokens, dim), frames <= 32;
    `residual_values` (batch, frames, tokens, heads, dim_head)."""
    return _TimeAttention.apply(x, residual_values, norm_weight, to_q, to_k, to_v, to_out, to_gates, mix_weight, mix_bias, k_gamma, inv_freq,
                                softclamp_value, belief)
