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
