"""Stand-in for `assoc-scan` (unpinned in pyproject.toml:30). Restates the
linear recurrence h_t = g_t * h_{t±1} + x_t that AssocScan computes (the
package's README definition); reverse=True scans from the last index.
Used at dreamer4.py:1594-1596 only."""
import torch
from torch.nn import Module

class AssocScan(Module):
    def __init__(self, reverse = False, use_accelerated = False):
        super().__init__()
        self.reverse = reverse

    def forward(self, gates, inputs, prev = None):
        n = gates.shape[-1]
        out = torch.empty_like(inputs)
        h = prev if prev is not None else torch.zeros_like(inputs[..., 0])
        order = range(n - 1, -1, -1) if self.reverse else range(n)
        for i in order:
            h = gates[..., i] * h + inputs[..., i]
            out[..., i] = h
        return out
