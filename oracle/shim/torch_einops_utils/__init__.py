"""Stand-in for `torch-einops-utils>=0.1.5` (pyproject.toml:48): the small
tensor helpers dreamer4.py imports at :53-72, restated from their names and
call sites.  PARITY UNPINNED against the real package (only masked_mean's
eps handling is numerically visible; with a non-empty mask it is
sum(t*mask)/count)."""
import torch
import torch.nn.functional as F
from torch.utils._pytree import tree_flatten, tree_unflatten, tree_map

def exists(v): return v is not None

def maybe(fn):
    def inner(t, *a, **k):
        if not exists(t): return None
        return fn(t, *a, **k)
    return inner

def pad_right_ndim_to(t, ndim):
    return t.reshape(*t.shape, *((1,) * (ndim - t.ndim))) if t.ndim < ndim else t

def align_dims_left(tensors):
    ndim = max(t.ndim for t in tensors)
    return tuple(pad_right_ndim_to(t, ndim) for t in tensors)

def pad_at_dim(t, pad, *, dim = -1, value = 0.):
    dims_from_right = (-dim - 1) if dim < 0 else (t.ndim - dim - 1)
    return F.pad(t, ((0, 0) * dims_from_right) + tuple(pad), value = value)

def pad_left_at_dim(t, pad, **kw): return pad_at_dim(t, (pad, 0), **kw)
def pad_right_at_dim(t, pad, **kw): return pad_at_dim(t, (0, pad), **kw)

def pad_right_at_dim_to(t, length, *, dim = -1, value = 0.):
    cur = t.shape[dim]
    return t if cur >= length else pad_right_at_dim(t, length - cur, dim = dim, value = value)

def lens_to_mask(lens, max_len = None):
    max_len = int(lens.amax().item()) if max_len is None else max_len
    return torch.arange(max_len, device = lens.device) < lens[..., None]

def shift_right(t, dim = 1, value = 0.):
    return pad_at_dim(t.narrow(dim, 0, t.shape[dim] - 1), (1, 0), dim = dim, value = value)

def masked_mean(t, mask = None, dim = None, eps = 1e-5):
    if not exists(mask):
        return t.mean(dim = dim) if exists(dim) else t.mean()
    mask = pad_right_ndim_to(mask, t.ndim).expand_as(t)
    if not exists(dim):
        return t[mask].mean() if mask.any() else t[mask].sum()
    num = (t * mask).sum(dim = dim)
    den = mask.sum(dim = dim)
    return num / den.clamp(min = eps)

def repeat_interleave_to_match(t, target):
    return t.repeat_interleave(target.shape[0] // t.shape[0], dim = 0)

def safe_stack(tensors, dim = 0):
    tensors = [t for t in tensors if exists(t)]
    return torch.stack(tensors, dim = dim) if len(tensors) > 0 else None

def safe_cat(tensors, dim = 0):
    tensors = [t for t in tensors if exists(t)]
    return torch.cat(tensors, dim = dim) if len(tensors) > 0 else None

def tree_flatten_with_inverse(tree):
    flat, spec = tree_flatten(tree)
    return flat, (lambda out: tree_unflatten(list(out), spec))

def tree_map_tensor(fn, tree):
    return tree_map(lambda t: fn(t) if torch.is_tensor(t) else t, tree)
