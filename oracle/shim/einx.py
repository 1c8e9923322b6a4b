"""Stand-in for `einx>=0.3.0` (pyproject.toml:32): only the named-axis
broadcasting elementwise ops the dynamics path calls (dreamer4.py:414, 542,
1236-1237, 1525, 1545, 1679, 7010-7173).  Semantics per the einx docs: every
operand is aligned to the output expression by axis name; with no '->' the
output is the operand expression that names every axis."""
import torch

def _tokens(expr, ndim):
    toks = expr.split()
    if '...' in toks:
        i = toks.index('...')
        n_ell = ndim - (len(toks) - 1)
        toks = toks[:i] + [f'_e{j}' for j in range(n_ell)] + toks[i + 1:]
    assert len(toks) == ndim, (expr, ndim)
    return toks

def _elementwise(op, pattern, *tensors):
    if '->' in pattern:
        ins, out = pattern.split('->')
    else:
        ins, out = pattern, None
    in_exprs = [e.strip() for e in ins.split(',')]
    assert len(in_exprs) == len(tensors)
    tensors = [t if torch.is_tensor(t) else torch.tensor(t) for t in tensors]
    in_toks = [_tokens(e, t.ndim) for e, t in zip(in_exprs, tensors)]
    if out is None:
        all_axes = set(a for toks in in_toks for a in toks if a != '1')
        cands = [toks for toks in in_toks if all_axes <= set(toks)]
        assert cands, f'no operand names all axes in {pattern}'
        out_toks = cands[0]
    else:
        # ellipsis in output takes the ellipsis dims of whichever operand has them
        n_ell = max([sum(a.startswith('_e') for a in toks) for toks in in_toks] + [0])
        toks = out.split()
        if '...' in toks:
            i = toks.index('...')
            toks = toks[:i] + [f'_e{j}' for j in range(n_ell)] + toks[i + 1:]
        out_toks = toks
    aligned = []
    for toks, t in zip(in_toks, tensors):
        # drop literal-1 axes, then permute/unsqueeze to out order
        keep = [i for i, a in enumerate(toks) if a != '1']
        t = t.reshape([t.shape[i] for i in keep])
        toks = [toks[i] for i in keep]
        perm = [toks.index(a) for a in out_toks if a in toks]
        t = t.permute(perm)
        shape_iter = iter(t.shape)
        view = [next(shape_iter) if a in toks else 1 for a in out_toks]
        aligned.append(t.reshape(view))
    return op(*aligned)

def add(p, *t): return _elementwise(lambda a, b: a + b, p, *t)
def multiply(p, *t): return _elementwise(lambda a, b: a * b, p, *t)
def equal(p, *t): return _elementwise(lambda a, b: a == b, p, *t)
def logical_and(p, *t): return _elementwise(lambda a, b: a & b, p, *t)
def greater_equal(p, *t): return _elementwise(lambda a, b: a >= b, p, *t)
def less(p, *t): return _elementwise(lambda a, b: a < b, p, *t)

def _off_path(*a, **k): raise NotImplementedError('einx op not on the imagination path')
dot = where = _off_path
