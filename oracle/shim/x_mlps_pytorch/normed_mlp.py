"""Normed MLP stand-in (see x_mlps_pytorch/__init__.py).  RECIPE selects the layer recipe — the real package cannot be
consulted here, so both plausible forms exist and every fixture records which one produced it:
  'pre_rms'     layer = Sequential(RMSNorm(d_in), Linear(d_in, d_out), activation)         (no activation on the last layer)
  'post_layer'  layer = Sequential(Linear(d_in, d_out), LayerNorm(d_out), activation); the last layer is a bare Linear
"""
from torch import nn
from torch.nn import Module, ModuleList

RECIPE = 'pre_rms'

class MLP(Module):
    def __init__(self, *dims, activation = nn.ReLU(), bias = True, activate_last = False):
        super().__init__()
        assert len(dims) > 1
        pairs = list(zip(dims[:-1], dims[1:]))
        layers = []
        for i, (d_in, d_out) in enumerate(pairs, start = 1):
            is_last = i == len(pairs)
            if RECIPE == 'pre_rms':
                mods = [nn.RMSNorm(d_in), nn.Linear(d_in, d_out, bias = bias)]
                if not is_last or activate_last:
                    mods.append(activation)
                layers.append(nn.Sequential(*mods))
            elif RECIPE == 'post_layer':
                lin = nn.Linear(d_in, d_out, bias = bias)
                layers.append(nn.Sequential(lin, nn.LayerNorm(d_out), activation) if (not is_last or activate_last) else lin)
            else:
                raise ValueError(RECIPE)
        self.layers = ModuleList(layers)

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x

def create_mlp(dim, depth, *, dim_in = None, dim_out = None, **kw):
    dims = (dim,) * (depth + 1)
    if dim_in is not None: dims = (dim_in, *dims)
    if dim_out is not None: dims = (*dims, dim_out)
    return MLP(*dims, **kw)
