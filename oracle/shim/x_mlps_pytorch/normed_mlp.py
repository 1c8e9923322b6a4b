from torch import nn
from torch.nn import Module, ModuleList

class MLP(Module):
    def __init__(self, *dims, activation = nn.ReLU(), bias = True, activate_last = False):
        super().__init__()
        assert len(dims) > 1
        pairs = list(zip(dims[:-1], dims[1:]))
        layers = []
        for i, (d_in, d_out) in enumerate(pairs, start = 1):
            is_last = i == len(pairs)
            mods = [nn.RMSNorm(d_in), nn.Linear(d_in, d_out, bias = bias)]
            if not is_last or activate_last:
                mods.append(activation)
            layers.append(nn.Sequential(*mods))
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
