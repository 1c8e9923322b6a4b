"""Ensemble(net, n): n independently initialised copies of `net` with stacked
parameters; forward_one(x, id) evaluates member `id` (dreamer4.py:5072, 6598).
ASSUMED: members are re-initialised copies (here: deep copies with
reset_parameters where available); parameter naming `params.<flat name>`."""
from copy import deepcopy
import torch
from torch import nn
from torch.func import functional_call

class Ensemble(nn.Module):
    def __init__(self, net, ensemble_size):
        super().__init__()
        self.net = [net]  # not registered; template only
        self.ensemble_size = ensemble_size
        members = []
        for _ in range(ensemble_size):
            m = deepcopy(net)
            for sub in m.modules():
                if hasattr(sub, 'reset_parameters'): sub.reset_parameters()
            members.append(m)
        self.names = [n for n, _ in net.named_parameters()]
        self.params = nn.ParameterList([
            nn.Parameter(torch.stack([dict(m.named_parameters())[n].detach() for m in members]))
            for n in self.names])

    def _member(self, i):
        return {n: p[i] for n, p in zip(self.names, self.params)}

    def forward_one(self, x, id):
        return functional_call(self.net[0], self._member(id), (x,))

    def forward(self, x):
        return torch.stack([self.forward_one(x, i) for i in range(self.ensemble_size)])
