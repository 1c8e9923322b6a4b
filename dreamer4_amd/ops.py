"""`torch.library` registration of the single-kernel entry points of the C-ABI (SURVEY.md 8b "C++ op library"): the ops
are visible to the PyTorch dispatcher as `torch.ops.d4hip.*`, carry fake (meta) implementations so that
`torch.compile(fullgraph=True)` traces through them without a graph break, and run the HIP kernels of libd4hip.so on the
current stream.  There is no CPU implementation: on a CPU tensor they raise (the product has no fallback path).

    y   = torch.ops.d4hip.rmsnorm(x, weight, eps)                       nn.RMSNorm                          (d4_rmsnorm)
    out = torch.ops.d4hip.linear(x, weight, bias, residual, flags, eps) Linear with the fused epilogues     (d4_gemm; flags: _lib.GEMM_*)
    v   = torch.ops.d4hip.hl_gauss_to_scalar(logits, centers)           HLGaussRewardEncoder.bins_to_scalar_value  D4:1088-1096
    ret = torch.ops.d4hip.gae(rewards, values, lens, is_truncated, terminals, gamma, lam)     calc_gae + masks  D4:1566-1600, 5943-5971

The stateful calls (rollout, learner) keep their engine handle and therefore stay methods of DynamicsWorldModel."""
from __future__ import annotations

import ctypes as C

import torch
from torch.library import custom_op

from dreamer4_amd import _lib


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and t.device.type != 'cuda':
            raise _lib.D4Error('d4hip ops run only on an MI355X (HIP) device: there is no CPU fallback')


@custom_op('d4hip::rmsnorm', mutates_args=())
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    _need_gpu(x, weight)
    lib = _lib.load()
    x2 = x.float().contiguous().view(-1, x.shape[-1])
    y = torch.empty_like(x2)
    _lib.check(lib.d4_rmsnorm(_lib.ptr(x2), x2.shape[1], _lib.ptr(weight.float().contiguous()), _lib.ptr(y), x2.shape[1], x2.shape[0], x2.shape[1], eps, _stream(x)))
    return y.view(x.shape)


@rmsnorm.register_fake
def _(x, weight, eps):
    return torch.empty_like(x, dtype=torch.float32)


@custom_op('d4hip::linear', mutates_args=())
def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None, residual: torch.Tensor | None, flags: int, eps: float) -> torch.Tensor:
    """x (..., K) @ weight (N, K)^T with the engine's fused epilogues (flags: GEMM_RMS_ROWSCALE | GEMM_SILU | GEMM_SWIGLU)."""
    _need_gpu(x, weight)
    lib = _lib.load()
    K, N = x.shape[-1], weight.shape[0]
    x2 = x.float().contiguous().view(-1, K)
    M = x2.shape[0]
    n_out = N // 2 if flags & _lib.GEMM_SWIGLU else N
    out = torch.empty(M, n_out, device=x.device)
    r2 = residual.float().contiguous().view(M, N) if residual is not None else None
    b = bias.float().contiguous() if bias is not None else None
    _lib.check(lib.d4_gemm(_lib.ptr(x2), K, _lib.ptr(weight.float().contiguous()), K, _lib.ptr(out), n_out, _lib.ptr(b), _lib.ptr(r2), N, M, N, K, flags, eps, _stream(x)))
    return out.view(*x.shape[:-1], n_out)


@linear.register_fake
def _(x, weight, bias, residual, flags, eps):
    n = weight.shape[0] // 2 if flags & _lib.GEMM_SWIGLU else weight.shape[0]
    return x.new_empty(*x.shape[:-1], n, dtype=torch.float32)


@custom_op('d4hip::hl_gauss_to_scalar', mutates_args=())
def hl_gauss_to_scalar(logits: torch.Tensor, centers: torch.Tensor) -> torch.Tensor:
    _need_gpu(logits, centers)
    lib = _lib.load()
    l2 = logits.float().contiguous().view(-1, logits.shape[-1])
    out = torch.empty(l2.shape[0], device=logits.device)
    _lib.check(lib.d4_hl_gauss_scalar(_lib.ptr(l2), l2.shape[1], _lib.ptr(centers.float().contiguous()), _lib.ptr(out), l2.shape[0], l2.shape[1], _stream(logits)))
    return out.view(logits.shape[:-1])


@hl_gauss_to_scalar.register_fake
def _(logits, centers):
    return logits.new_empty(logits.shape[:-1], dtype=torch.float32)


@custom_op('d4hip::gae', mutates_args=())
def gae(rewards: torch.Tensor, values: torch.Tensor, lens: torch.Tensor | None, is_truncated: torch.Tensor | None, terminals: torch.Tensor | None,
        gamma: float, lam: float) -> torch.Tensor:
    _need_gpu(rewards, values)
    lib = _lib.load()
    r, v = rewards.float().contiguous(), values.float().contiguous()
    out = torch.empty_like(r)
    ln = lens.long().contiguous() if lens is not None else None
    tr = is_truncated.to(torch.uint8).contiguous() if is_truncated is not None else None
    te = terminals.to(torch.uint8).contiguous() if terminals is not None else None
    _lib.check(lib.d4_gae(_lib.ptr(r), _lib.ptr(v), _lib.ptr(ln), _lib.ptr(tr), _lib.ptr(te), gamma, lam, r.shape[0], r.shape[1], _lib.ptr(out), _stream(rewards)))
    return out


@gae.register_fake
def _(rewards, values, lens, is_truncated, terminals, gamma, lam):
    return torch.empty_like(rewards, dtype=torch.float32)


# ------------------------------------------------------------------------------------------------ autograd for rmsnorm / linear
@custom_op('d4hip::rmsnorm_backward', mutates_args=())
def rmsnorm_backward(x: torch.Tensor, dy: torch.Tensor, weight: torch.Tensor, eps: float) -> tuple[torch.Tensor, torch.Tensor]:
    _need_gpu(x, dy, weight)
    lib = _lib.load()
    D = x.shape[-1]
    x2, dy2 = x.float().contiguous().view(-1, D), dy.float().contiguous().view(-1, D)
    dx, dw, scratch = torch.empty_like(x2), torch.empty(D, device=x.device), torch.empty_like(x2)
    _lib.check(lib.d4_rmsnorm_backward(_lib.ptr(x2), _lib.ptr(dy2), _lib.ptr(weight.float().contiguous()), _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(scratch),
                                       x2.shape[0], D, eps, _stream(x)))
    return dx.view(x.shape), dw


@rmsnorm_backward.register_fake
def _(x, dy, weight, eps):
    return torch.empty_like(x, dtype=torch.float32), torch.empty_like(weight, dtype=torch.float32)


def _rmsnorm_setup(ctx, inputs, output):
    x, weight, eps = inputs
    ctx.save_for_backward(x, weight)
    ctx.eps = eps


def _rmsnorm_bwd(ctx, dy):
    x, weight = ctx.saved_tensors
    dx, dw = torch.ops.d4hip.rmsnorm_backward(x, dy, weight, ctx.eps)
    return dx, dw, None


rmsnorm.register_autograd(_rmsnorm_bwd, setup_context=_rmsnorm_setup)


@custom_op('d4hip::linear_backward', mutates_args=())
def linear_backward(x: torch.Tensor, dy: torch.Tensor, weight: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """Gradients of y = x @ weight^T: dx = dy @ weight, dweight = dy^T @ x (d4_gemm with transposed operands)."""
    _need_gpu(x, dy, weight)
    lib = _lib.load()
    K, N = x.shape[-1], weight.shape[0]
    x2, dy2, w = x.float().contiguous().view(-1, K), dy.float().contiguous().view(-1, N), weight.float().contiguous()
    M = x2.shape[0]
    n_out = N
    if N % 4:            # a head's last Linear (1 or a few outputs): zero-pad the output width to the kernels' 4-float granularity
        pad = 4 - N % 4
        dy2, w, N = torch.nn.functional.pad(dy2, (0, pad)), torch.nn.functional.pad(w, (0, 0, 0, pad)), N + pad
    dx, dw = torch.empty_like(x2), torch.empty_like(w)
    # dx[m][k] = sum_n dy[m][n] W[n][k]: "W(k, n)" read from W[n * K + k] -> TRANS_B
    _lib.check(lib.d4_gemm(_lib.ptr(dy2), N, _lib.ptr(w), K, _lib.ptr(dx), K, None, None, 0, M, K, N, _lib.GEMM_TRANS_B, 0., _stream(x)))
    # dw[n][k] = sum_m dy[m][n] x[m][k]: both operands indexed by the contraction first -> the weight-gradient kernel (d4_gemm_tn: operands
    # straight into the MFMA layout, rows split into slices) or, for widths that are not multiples of 4, d4_gemm with TRANS_A | TRANS_B
    if N % 4 == 0 and K % 4 == 0 and N >= 4 and K >= 4:
        part_floats = min(8 * N * K, 8 << 20) if M >= 512 else 0
        part = torch.empty(part_floats, device=x.device) if part_floats else None
        _lib.check(lib.d4_gemm_tn(_lib.ptr(dy2), N, _lib.ptr(x2), K, _lib.ptr(dw), K, N, K, M, _lib.ptr(part), part_floats, 0, 0, _stream(x)))
    else:
        _lib.check(lib.d4_gemm(_lib.ptr(dy2), N, _lib.ptr(x2), K, _lib.ptr(dw), K, None, None, 0, N, K, M, _lib.GEMM_TRANS_A | _lib.GEMM_TRANS_B, 0., _stream(x)))
    return dx.view(x.shape), (dw[:n_out].clone() if n_out != N else dw)


@linear_backward.register_fake
def _(x, dy, weight):
    return torch.empty_like(x, dtype=torch.float32), torch.empty_like(weight, dtype=torch.float32)


def _linear_setup(ctx, inputs, output):
    x, weight, bias, residual, flags, eps = inputs
    if flags != 0:
        raise _lib.D4Error('d4hip::linear is differentiable only without fused epilogue flags (use the block operators for the fused forms)')
    ctx.save_for_backward(x, weight)
    ctx.has_bias, ctx.has_res = bias is not None, residual is not None


def _linear_bwd(ctx, dy):
    x, weight = ctx.saved_tensors
    dx, dw = torch.ops.d4hip.linear_backward(x, dy, weight)
    db = dy.reshape(-1, dy.shape[-1]).sum(0) if ctx.has_bias else None
    return dx, dw, db, (dy if ctx.has_res else None), None, None


linear.register_autograd(_linear_bwd, setup_context=_linear_setup)


# ------------------------------------------------------------------------------------------------ differentiable trunk blocks
# FeedForward (D4:2079-2116) and Attention (D4:1968-2075: within a frame / along time / over a context — the last one covers AttentionPool,
# the special-token cross attention and LearnedQueriesAttentionPool) as dispatcher-visible ops with registered autograd.  Every forward
# also returns the uint8 workspace it ran in (empty when D4_TRUNK_SAVE_FORWARD=0): the backward op runs on it without recomputing the
# forward (`d4_*_backward_saved`), or recomputes it when the workspace is empty.  Parameters and gradients in the reference's layout.
def _save_ws():
    import os
    return os.environ.get('D4_TRUNK_SAVE_FORWARD', '1') != '0'


def _ws_new(nbytes, device):
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    return ws


def _ws_ptr(ws):
    base = ws.data_ptr()
    return C.c_void_p(base + (-base) % 256)


def _f32c(*ts):
    out = []
    for t in ts:
        if t is None:
            out.append(None)
            continue
        if t.dtype != torch.float32:
            raise _lib.D4Error('d4hip trunk blocks are fp32')
        out.append(t.contiguous())
    return out


def _opt(t):                      # custom ops cannot return None: an absent gradient is an empty tensor
    return None if t is None or t.numel() == 0 else t


@custom_op('d4hip::swiglu_ff', mutates_args=())
def swiglu_ff(x: torch.Tensor, norm_w: torch.Tensor, w_in: torch.Tensor, b_in: torch.Tensor, w_out: torch.Tensor,
              b_out: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    _need_gpu(x)
    x, norm_w, w_in, b_in, w_out, b_out = _f32c(x, norm_w, w_in, b_in, w_out, b_out)
    D, inner = x.shape[-1], w_out.shape[1]
    assert w_in.shape == (2 * inner, D) and w_out.shape == (D, inner) and b_in.shape == (2 * inner,) and b_out.shape == (D,)
    rows = x.numel() // D
    lib = _lib.load()
    nbytes = lib.d4_ff_workspace_bytes(rows, D, inner)
    ws = _ws_new(nbytes, x.device)
    y = torch.empty_like(x)
    _lib.check(lib.d4_ff_forward(_lib.ptr(x), _lib.ptr(norm_w), _lib.ptr(w_in), _lib.ptr(b_in), _lib.ptr(w_out), _lib.ptr(b_out),
                                 rows, D, inner, _lib.ptr(y), _ws_ptr(ws), nbytes, _stream(x)))
    return y, (ws if _save_ws() else ws.new_empty(0))


@swiglu_ff.register_fake
def _(x, norm_w, w_in, b_in, w_out, b_out):
    return torch.empty_like(x), x.new_empty(0, dtype=torch.uint8)


@custom_op('d4hip::swiglu_ff_backward', mutates_args=())
def swiglu_ff_backward(x: torch.Tensor, dy: torch.Tensor, norm_w: torch.Tensor, w_in: torch.Tensor, b_in: torch.Tensor, w_out: torch.Tensor,
                       ws: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    x, dy, norm_w, w_in, b_in, w_out = _f32c(x, dy, norm_w, w_in, b_in, w_out)
    D, inner = x.shape[-1], w_out.shape[1]
    rows = x.numel() // D
    lib = _lib.load()
    nbytes = lib.d4_ff_workspace_bytes(rows, D, inner)
    fn = lib.d4_ff_backward_saved
    if ws.numel() == 0:
        ws, fn = _ws_new(nbytes, x.device), lib.d4_ff_backward
    dx, dn, dwi, dbi, dwo = torch.empty_like(x), torch.empty_like(norm_w), torch.empty_like(w_in), torch.empty_like(b_in), torch.empty_like(w_out)
    dbo = torch.empty(D, device=x.device)
    _lib.check(fn(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(norm_w), _lib.ptr(w_in), _lib.ptr(b_in), _lib.ptr(w_out), rows, D, inner,
                  _lib.ptr(dx), _lib.ptr(dn), _lib.ptr(dwi), _lib.ptr(dbi), _lib.ptr(dwo), _lib.ptr(dbo), _ws_ptr(ws), nbytes, _stream(x)))
    return dx, dn, dwi, dbi, dwo, dbo


@swiglu_ff_backward.register_fake
def _(x, dy, norm_w, w_in, b_in, w_out, ws):
    e = torch.empty_like
    return e(x), e(norm_w), e(w_in), e(b_in), e(w_out), x.new_empty(x.shape[-1])


def _ff_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs[:5], output[1])


def _ff_bwd(ctx, dy, _dws):
    x, norm_w, w_in, b_in, w_out, ws = ctx.saved_tensors
    return torch.ops.d4hip.swiglu_ff_backward(x, dy, norm_w, w_in, b_in, w_out, ws)


swiglu_ff.register_autograd(_ff_bwd, setup_context=_ff_setup)


def _attn_self(time_form, x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq, softclamp, num_special, belief):
    _need_gpu(x)
    x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq = _f32c(x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq)
    heads, dh = gamma.shape
    lib = _lib.load()
    y = torch.empty_like(x)
    P = _lib.ptr
    if time_form:
        B, T, S, D = x.shape
        assert wq.shape == (heads * dh, D) and wo.shape == (D, heads * dh) and wg.shape == (heads, D) and inv_freq.shape == (dh // 2,)
        assert rv is None or rv.shape == (B, T, S, heads, dh)
        nbytes = lib.d4_time_attn_workspace_bytes(B, T, S, D, heads, dh)
        ws = _ws_new(nbytes, x.device)
        _lib.check(lib.d4_time_attn_forward(P(x), P(rv), P(norm_w), P(wq), P(wk), P(wv), P(wo), P(wg), P(wm), P(bm), P(gamma), P(inv_freq), B, T, S, D, heads, dh,
                                            float(softclamp), int(belief), P(y), _ws_ptr(ws), nbytes, _stream(x)))
    else:
        F_, S, D = x.shape
        assert wq.shape == (heads * dh, D) and wo.shape == (D, heads * dh) and wg.shape == (heads, D)
        assert rv is None or rv.shape == (F_, S, heads, dh)
        nbytes = lib.d4_attn_workspace_bytes(F_, S, D, heads, dh)
        ws = _ws_new(nbytes, x.device)
        _lib.check(lib.d4_space_attn_forward(P(x), P(rv), P(norm_w), P(wq), P(wk), P(wv), P(wo), P(wg), P(wm), P(bm), P(gamma), F_, S, D, heads, dh,
                                             float(softclamp), int(num_special), int(belief), P(y), _ws_ptr(ws), nbytes, _stream(x)))
    return y, (ws if _save_ws() else ws.new_empty(0))


def _attn_self_backward(time_form, x, rv, dy, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq, softclamp, num_special, belief, ws):
    x, rv, dy, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq = _f32c(x, rv, dy, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq)
    heads, dh = gamma.shape
    lib = _lib.load()
    e = torch.empty_like
    dx, dn, dq, dk, dv, do, dg, dgam = e(x), e(norm_w), e(wq), e(wk), e(wv), e(wo), e(wg), e(gamma)
    drv, dwm, dbm = (e(rv), e(wm), e(bm)) if rv is not None else (None, None, None)
    P = _lib.ptr
    saved = ws.numel() > 0
    if time_form:
        B, T, S, D = x.shape
        nbytes = lib.d4_time_attn_workspace_bytes(B, T, S, D, heads, dh)
        if not saved:
            ws = _ws_new(nbytes, x.device)
        fn = lib.d4_time_attn_backward_saved if saved else lib.d4_time_attn_backward
        _lib.check(fn(P(x), P(rv), P(dy), P(norm_w), P(wq), P(wk), P(wv), P(wo), P(wg), P(wm), P(bm), P(gamma), P(inv_freq), B, T, S, D, heads, dh,
                      float(softclamp), int(belief), P(dx), P(drv), P(dn), P(dq), P(dk), P(dv), P(do), P(dg), P(dwm), P(dbm), P(dgam),
                      _ws_ptr(ws), nbytes, _stream(x)))
    else:
        F_, S, D = x.shape
        nbytes = lib.d4_attn_workspace_bytes(F_, S, D, heads, dh)
        if not saved:
            ws = _ws_new(nbytes, x.device)
        fn = lib.d4_space_attn_backward_saved if saved else lib.d4_space_attn_backward
        _lib.check(fn(P(x), P(rv), P(dy), P(norm_w), P(wq), P(wk), P(wv), P(wo), P(wg), P(wm), P(bm), P(gamma), F_, S, D, heads, dh,
                      float(softclamp), int(num_special), int(belief), P(dx), P(drv), P(dn), P(dq), P(dk), P(dv), P(do), P(dg), P(dwm), P(dbm), P(dgam),
                      _ws_ptr(ws), nbytes, _stream(x)))
    z = lambda: x.new_empty(0)                 # (fresh tensors: the outputs of an op may not alias each other)
    return dx, (drv if drv is not None else z()), dn, dq, dk, dv, do, dg, (dwm if dwm is not None else z()), (dbm if dbm is not None else z()), dgam


_T = torch.Tensor
_G11 = tuple[_T, _T, _T, _T, _T, _T, _T, _T, _T, _T, _T]


@custom_op('d4hip::attn_block_space', mutates_args=())
def attn_block_space(x: _T, rv: _T | None, norm_w: _T, wq: _T, wk: _T, wv: _T, wo: _T, wg: _T, wm: _T | None, bm: _T | None, gamma: _T,
                     softclamp: float, num_special: int, belief: bool) -> tuple[_T, _T]:
    """Attention.forward within each frame (D4:1968-2075): x (frames, tokens, dim); rv = residual values (frames, tokens, heads, dim_head)."""
    return _attn_self(False, x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, None, softclamp, num_special, belief)


@attn_block_space.register_fake
def _(x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, softclamp, num_special, belief):
    return torch.empty_like(x), x.new_empty(0, dtype=torch.uint8)


@custom_op('d4hip::attn_block_space_backward', mutates_args=())
def attn_block_space_backward(x: _T, rv: _T | None, dy: _T, norm_w: _T, wq: _T, wk: _T, wv: _T, wo: _T, wg: _T, wm: _T | None, bm: _T | None, gamma: _T,
                              softclamp: float, num_special: int, belief: bool, ws: _T) -> _G11:
    return _attn_self_backward(False, x, rv, dy, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, None, softclamp, num_special, belief, ws)


def _fake_g11(x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma):
    e = torch.empty_like
    z = lambda: x.new_empty(0)
    return e(x), (e(rv) if rv is not None else z()), e(norm_w), e(wq), e(wk), e(wv), e(wo), e(wg), (e(wm) if wm is not None else z()), (e(bm) if bm is not None else z()), e(gamma)


@attn_block_space_backward.register_fake
def _(x, rv, dy, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, softclamp, num_special, belief, ws):
    return _fake_g11(x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma)


def _space_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs[:11], output[1])
    ctx.cfg = inputs[11:14]


def _space_bwd(ctx, dy, _dws):
    *t, ws = ctx.saved_tensors
    x, rv = t[0], t[1]
    g = torch.ops.d4hip.attn_block_space_backward(x, rv, dy, *t[2:], *ctx.cfg, ws)
    return (g[0], _opt(g[1]), *g[2:8], _opt(g[8]), _opt(g[9]), g[10], None, None, None)


attn_block_space.register_autograd(_space_bwd, setup_context=_space_setup)


@custom_op('d4hip::attn_block_time', mutates_args=())
def attn_block_time(x: _T, rv: _T | None, norm_w: _T, wq: _T, wk: _T, wv: _T, wo: _T, wg: _T, wm: _T | None, bm: _T | None, gamma: _T, inv_freq: _T,
                    softclamp: float, belief: bool) -> tuple[_T, _T]:
    """The trunk's time layers (D4:3176-3215): causal attention along time per token column with rotary positions, training form (no KV
    cache): x (batch, frames, tokens, dim)."""
    return _attn_self(True, x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq, softclamp, 0, belief)


@attn_block_time.register_fake
def _(x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq, softclamp, belief):
    return torch.empty_like(x), x.new_empty(0, dtype=torch.uint8)


@custom_op('d4hip::attn_block_time_backward', mutates_args=())
def attn_block_time_backward(x: _T, rv: _T | None, dy: _T, norm_w: _T, wq: _T, wk: _T, wv: _T, wo: _T, wg: _T, wm: _T | None, bm: _T | None, gamma: _T,
                             inv_freq: _T, softclamp: float, belief: bool, ws: _T) -> _G11:
    return _attn_self_backward(True, x, rv, dy, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq, softclamp, 0, belief, ws)


@attn_block_time_backward.register_fake
def _(x, rv, dy, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma, inv_freq, softclamp, belief, ws):
    return _fake_g11(x, rv, norm_w, wq, wk, wv, wo, wg, wm, bm, gamma)


def _time_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs[:12], output[1])
    ctx.cfg = inputs[12:14]


def _time_bwd(ctx, dy, _dws):
    *t, ws = ctx.saved_tensors
    g = torch.ops.d4hip.attn_block_time_backward(t[0], t[1], dy, *t[2:12], *ctx.cfg, ws)
    return (g[0], _opt(g[1]), *g[2:8], _opt(g[8]), _opt(g[9]), g[10], None, None, None)


attn_block_time.register_autograd(_time_bwd, setup_context=_time_setup)


@custom_op('d4hip::attn_block_cross', mutates_args=())
def attn_block_cross(q_tokens: _T, context: _T, norm_w: _T, norm_ctx_w: _T | None, wq: _T, wk: _T, wv: _T, wo: _T, wg: _T, gamma: _T, item_major: bool,
                     softclamp: float) -> tuple[_T, _T]:
    """Attention.forward with a context (D4:1968-2075) = AttentionPool (context = the item-major stack of layer hiddens, D4:2143-2177), the
    special-token cross attention (D4:3227-3234) and LearnedQueriesAttentionPool (D4:2179-2210): q_tokens (groups, nq, dim)."""
    _need_gpu(q_tokens)
    q_tokens, context, norm_w, norm_ctx_w, wq, wk, wv, wo, wg, gamma = _f32c(q_tokens, context, norm_w, norm_ctx_w, wq, wk, wv, wo, wg, gamma)
    G, nq, D = q_tokens.shape
    nk, Dc = (context.shape[0], context.shape[2]) if item_major else (context.shape[1], context.shape[2])
    assert (context.shape[1] if item_major else context.shape[0]) == G, 'context groups do not match the queries'
    heads, dh = gamma.shape
    assert wq.shape == (heads * dh, D) and wk.shape == (heads * dh, Dc) and wv.shape == (heads * dh, Dc) and wo.shape == (D, heads * dh)
    lib = _lib.load()
    nbytes = lib.d4_cross_attn_workspace_bytes(G, nq, nk, D, Dc, heads, dh)
    ws = _ws_new(nbytes, q_tokens.device)
    y = torch.empty_like(q_tokens)
    P = _lib.ptr
    _lib.check(lib.d4_cross_attn_forward(P(q_tokens), P(context), P(norm_w), P(norm_ctx_w), P(wq), P(wk), P(wv), P(wo), P(wg), P(gamma), G, nq, nk,
                                         int(item_major), D, Dc, heads, dh, float(softclamp), P(y), _ws_ptr(ws), nbytes, _stream(q_tokens)))
    return y, (ws if _save_ws() else ws.new_empty(0))


@attn_block_cross.register_fake
def _(q_tokens, context, norm_w, norm_ctx_w, wq, wk, wv, wo, wg, gamma, item_major, softclamp):
    return torch.empty_like(q_tokens), q_tokens.new_empty(0, dtype=torch.uint8)


_G10 = tuple[_T, _T, _T, _T, _T, _T, _T, _T, _T, _T]


@custom_op('d4hip::attn_block_cross_backward', mutates_args=())
def attn_block_cross_backward(q_tokens: _T, context: _T, dy: _T, norm_w: _T, norm_ctx_w: _T | None, wq: _T, wk: _T, wv: _T, wo: _T, wg: _T, gamma: _T,
                              item_major: bool, softclamp: float, ws: _T) -> _G10:
    q_tokens, context, dy, norm_w, norm_ctx_w, wq, wk, wv, wo, wg, gamma = _f32c(q_tokens, context, dy, norm_w, norm_ctx_w, wq, wk, wv, wo, wg, gamma)
    G, nq, D = q_tokens.shape
    nk, Dc = (context.shape[0], context.shape[2]) if item_major else (context.shape[1], context.shape[2])
    heads, dh = gamma.shape
    lib = _lib.load()
    nbytes = lib.d4_cross_attn_workspace_bytes(G, nq, nk, D, Dc, heads, dh)
    saved = ws.numel() > 0
    if not saved:
        ws = _ws_new(nbytes, q_tokens.device)
    fn = lib.d4_cross_attn_backward_saved if saved else lib.d4_cross_attn_backward
    e = torch.empty_like
    dq_t, dc, dn, dq, dk, dv, do, dg, dgam = e(q_tokens), e(context), e(norm_w), e(wq), e(wk), e(wv), e(wo), e(wg), e(gamma)
    dnc = e(norm_ctx_w) if norm_ctx_w is not None else None
    P = _lib.ptr
    _lib.check(fn(P(q_tokens), P(context), P(dy), P(norm_w), P(norm_ctx_w), P(wq), P(wk), P(wv), P(wo), P(wg), P(gamma), G, nq, nk, int(item_major), D, Dc,
                  heads, dh, float(softclamp), P(dq_t), P(dc), P(dn), P(dnc), P(dq), P(dk), P(dv), P(do), P(dg), P(dgam), _ws_ptr(ws), nbytes, _stream(q_tokens)))
    return dq_t, dc, dn, (dnc if dnc is not None else q_tokens.new_empty(0)), dq, dk, dv, do, dg, dgam


@attn_block_cross_backward.register_fake
def _(q_tokens, context, dy, norm_w, norm_ctx_w, wq, wk, wv, wo, wg, gamma, item_major, softclamp, ws):
    e = torch.empty_like
    return e(q_tokens), e(context), e(norm_w), (e(norm_ctx_w) if norm_ctx_w is not None else q_tokens.new_empty(0)), e(wq), e(wk), e(wv), e(wo), e(wg), e(gamma)


def _cross_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs[:10], output[1])
    ctx.cfg = inputs[10:12]


def _cross_bwd(ctx, dy, _dws):
    *t, ws = ctx.saved_tensors
    g = torch.ops.d4hip.attn_block_cross_backward(t[0], t[1], dy, *t[2:10], *ctx.cfg, ws)
    return (g[0], g[1], g[2], _opt(g[3]), *g[4:10], None, None)


attn_block_cross.register_autograd(_cross_bwd, setup_context=_cross_setup)


@custom_op('d4hip::flow_euler_step', mutates_args=())
def flow_euler_step(x: torch.Tensor, pred: torch.Tensor, one_minus_t: float, dt: float) -> torch.Tensor:
    """The x-space shortcut / Euler update of the denoising loop (D4:6567-6580): x + (pred - x) / (1 - t) * dt, as a new tensor."""
    _need_gpu(x, pred)
    out = x.float().contiguous().clone()
    _lib.check(_lib.load().d4_euler_step(_lib.ptr(out), _lib.ptr(pred.float().contiguous()), out.numel(), one_minus_t, dt, _stream(x)))
    return out


@flow_euler_step.register_fake
def _(x, pred, one_minus_t, dt):
    return torch.empty_like(x, dtype=torch.float32)


# ------------------------------------------------------------------------------------------------ sampling / value loss (stateless)
@custom_op('d4hip::categorical_sample_logp', mutates_args=())
def categorical_sample_logp(logits: torch.Tensor, uniform: torch.Tensor, action_sizes: list[int], temperature: float) -> tuple[torch.Tensor, torch.Tensor]:
    """MultiCategorical.sample + log_prob of the sample (D4:485-497, 1374-1376, 1422-1423): Gumbel-max per action type from injected uniforms of the
    logits' shape; returns (actions int64 [..., na], log_probs [..., na])."""
    _need_gpu(logits, uniform)
    lib = _lib.load()
    l2 = logits.float().contiguous().view(-1, logits.shape[-1])
    u2 = uniform.float().contiguous().view(-1, uniform.shape[-1])
    na = len(action_sizes)
    assert sum(action_sizes) == l2.shape[1] == u2.shape[1] and na >= 1
    sizes = torch.tensor(action_sizes, dtype=torch.int32, device=logits.device)
    acts = torch.empty(l2.shape[0], na, dtype=torch.long, device=logits.device)
    lps = torch.empty(l2.shape[0], na, device=logits.device)
    _lib.check(lib.d4_categorical_sample_logp(_lib.ptr(l2), l2.shape[1], _lib.ptr(u2), u2.shape[1], _lib.ptr(sizes), l2.shape[0], na, temperature,
                                              _lib.ptr(acts), _lib.ptr(lps), _stream(logits)))
    return acts.view(*logits.shape[:-1], na), lps.view(*logits.shape[:-1], na)


@categorical_sample_logp.register_fake
def _(logits, uniform, action_sizes, temperature):
    na = len(action_sizes)
    return logits.new_empty(*logits.shape[:-1], na, dtype=torch.long), logits.new_empty(*logits.shape[:-1], na, dtype=torch.float32)


@custom_op('d4hip::hl_gauss_ce', mutates_args=())
def hl_gauss_ce(logits: torch.Tensor, targets: torch.Tensor, mask: torch.Tensor | None, support: torch.Tensor, vmin: float, vmax: float, sigma: float,
                eps: float, two_hot: bool) -> tuple[torch.Tensor, torch.Tensor]:
    """The value branch's loss (D4:6254-6295): cross entropy of `logits` against the HL-Gauss (support = bin edges) or two-hot (support = bin values)
    encoding of `targets`, mean over the masked rows.  Returns (loss, d loss / d logits); differentiable in `logits` through the second output."""
    _need_gpu(logits, targets, support)
    lib = _lib.load()
    l2 = logits.float().contiguous().view(-1, logits.shape[-1])
    t1 = targets.float().contiguous().view(-1)
    m1 = mask.float().contiguous().view(-1) if mask is not None else None
    sup = support.float().contiguous()
    rows, bins = l2.shape
    assert t1.numel() == rows and sup.numel() == (bins if two_hot else bins + 1)
    loss = torch.empty(1, device=logits.device)
    dl = torch.empty_like(l2)
    scratch = torch.empty(2 * rows + 64, device=logits.device)
    _lib.check(lib.d4_hl_gauss_ce(_lib.ptr(l2), bins, _lib.ptr(t1), _lib.ptr(m1), _lib.ptr(sup), rows, bins, vmin, vmax, sigma, eps, int(two_hot),
                                  _lib.ptr(loss), _lib.ptr(dl), _lib.ptr(scratch), _stream(logits)))
    return loss.view(()), dl.view(logits.shape)


@hl_gauss_ce.register_fake
def _(logits, targets, mask, support, vmin, vmax, sigma, eps, two_hot):
    return logits.new_empty((), dtype=torch.float32), torch.empty_like(logits, dtype=torch.float32)


def _hl_ce_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _hl_ce_bwd(ctx, d_loss, _d_dl):
    (dl,) = ctx.saved_tensors
    return dl * d_loss, None, None, None, None, None, None, None, None


hl_gauss_ce.register_autograd(_hl_ce_bwd, setup_context=_hl_ce_setup)


@custom_op('d4hip::ppo_policy_loss', mutates_args=())
def ppo_policy_loss(logits: torch.Tensor, actions: torch.Tensor, old_log_probs: torch.Tensor, advantages: torch.Tensor, mask: torch.Tensor | None,
                    action_sizes: list[int], objective: int, eps_clip: float, entropy_weight: float) -> tuple[torch.Tensor, torch.Tensor]:
    """The policy branch's loss for discrete actions (D4:6077-6242), stateless: PPO clipped surrogate (objective 0) or SPO (1) of the stored actions'
    joint log-prob against the behaviour log-probs, minus entropy_weight x entropy, mean over the masked rows; the advantages are taken as given.
    logits (..., sum(action_sizes)), actions / old_log_probs (..., na), advantages / mask (...).  Returns (loss, d loss / d logits); differentiable
    in `logits` through the second output (d4_ppo_policy_loss: the fused forward + backward kernel the learner runs)."""
    _need_gpu(logits, actions, old_log_probs, advantages)
    lib = _lib.load()
    l2 = logits.float().contiguous().view(-1, logits.shape[-1])
    na, total = len(action_sizes), sum(action_sizes)
    rows = l2.shape[0]
    a2 = actions.long().contiguous().view(rows, na)
    o2 = old_log_probs.float().contiguous().view(rows, na)
    adv = advantages.float().contiguous().view(rows)
    m1 = mask.float().contiguous().view(rows) if mask is not None else None
    assert total == l2.shape[1] and na >= 1
    sizes = torch.tensor(action_sizes, dtype=torch.int32, device=logits.device)
    loss = torch.empty(1, device=logits.device)
    dl = torch.empty_like(l2)
    scratch = torch.empty(5 * rows + 64, device=logits.device)
    _lib.check(lib.d4_ppo_policy_loss(_lib.ptr(l2), total, _lib.ptr(a2), _lib.ptr(o2), _lib.ptr(adv), _lib.ptr(m1), _lib.ptr(sizes), rows, na, total, objective,
                                      eps_clip, entropy_weight, _lib.ptr(loss), _lib.ptr(dl), _lib.ptr(scratch), _stream(logits)))
    return loss.view(()), dl.view(logits.shape)


@ppo_policy_loss.register_fake
def _(logits, actions, old_log_probs, advantages, mask, action_sizes, objective, eps_clip, entropy_weight):
    return logits.new_empty((), dtype=torch.float32), torch.empty_like(logits, dtype=torch.float32)


def _ppo_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _ppo_bwd(ctx, d_loss, _d_dl):
    (dl,) = ctx.saved_tensors
    return dl * d_loss, None, None, None, None, None, None, None, None


ppo_policy_loss.register_autograd(_ppo_bwd, setup_context=_ppo_setup)


def attn_pool(x: torch.Tensor, hiddens: torch.Tensor, norm_w: torch.Tensor, norm_ctx_w: torch.Tensor, wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor,
              wo: torch.Tensor, wg: torch.Tensor, gamma: torch.Tensor) -> torch.Tensor:
    """Residual(AttentionPool) (D4:2143-2177 + 1869) under its reference name: every token row of x (rows, dim) attends, as ONE query, over its own
    row in each of the L layer hiddens (L, rows, dim) (4 pool heads x 64), and the result is added to x.  A thin composition of the dispatcher op
    `torch.ops.d4hip.attn_block_cross` (context item-major; forward + registered backward: d4_cross_attn_forward / _backward) — differentiable in
    x, the hiddens and every weight, traceable under torch.compile."""
    assert x.ndim == 2 and hiddens.ndim == 3 and hiddens.shape[1:] == x.shape
    out = torch.ops.d4hip.attn_block_cross(x[:, None, :], hiddens, norm_w, norm_ctx_w, wq, wk, wv, wo, wg, gamma, True, 0.)[0]
    return x + out[:, 0, :]
