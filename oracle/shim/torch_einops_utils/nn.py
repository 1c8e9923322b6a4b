"""`Sequential` that drops None members (and renumbers), `Identity` that ignores extra args."""
from torch import nn

class Identity(nn.Module):
    def forward(self, x, *a, **k): return x

class Sequential(nn.Sequential):
    def __init__(self, *mods):
        super().__init__(*[m for m in mods if m is not None])
