"""Stand-in for `x-mlps-pytorch>=0.3.1` (pyproject.toml:50).  RECIPE ASSUMED,
PARITY UNPINNED: create_mlp(dim, depth, dim_in, dim_out) builds widths
(dim_in, dim x (depth+1), dim_out); the normed variant (normed_mlp.py) is
layer = [RMSNorm(d_in) -> Linear(d_in, d_out, bias) -> activation], the
activation omitted on the last layer.  Call sites: dreamer4.py:4950, 5083,
5095."""
from x_mlps_pytorch.normed_mlp import MLP, create_mlp
