"""Stand-in for `hl-gauss-pytorch` (unpinned, pyproject.toml:36).  Restates the
HL-Gauss transform published in Farebrother et al. 2024 ("Stop Regressing",
arXiv:2403.03950, listing 1): num_bins+1 uniform edges on [min,max],
probs = diff(erf((edges - y) / (sqrt(2) sigma))) / (cdf[-1]-cdf[0]),
scalar = sum(probs * bin_centres).  sigma = sigma_to_bin_ratio * bin_width.
Call sites: dreamer4.py:1059-1105.  PARITY UNPINNED against the real package."""
import math
import torch
from torch.nn import Module

class HLGaussLoss(Module):
    def __init__(self, min_value, max_value, num_bins, sigma = None, sigma_to_bin_ratio = None,
                 eps = 1e-10, clamp_to_range = False, min_max_value_on_bin_center = False):
        super().__init__()
        assert not min_max_value_on_bin_center, 'not restated'
        self.eps = eps
        self.min_value, self.max_value, self.num_bins = min_value, max_value, num_bins
        self.clamp_to_range = clamp_to_range
        support = torch.linspace(min_value, max_value, num_bins + 1).float()
        bin_size = (max_value - min_value) / num_bins
        if sigma is None:
            sigma = (sigma_to_bin_ratio if sigma_to_bin_ratio is not None else 2.) * bin_size
        self.sigma = sigma
        self.sigma_times_sqrt_two = math.sqrt(2.) * sigma
        self.register_buffer('support', support, persistent = False)
        self.register_buffer('centers', (support[:-1] + support[1:]) / 2, persistent = False)

    def transform_from_probs(self, probs):
        return (probs * self.centers).sum(dim = -1)

    def transform_from_logits(self, logits):
        return self.transform_from_probs(logits.softmax(dim = -1))

    def transform_to_probs(self, target, eps = None):
        eps = self.eps if eps is None else eps
        if self.clamp_to_range:
            target = target.clamp(self.min_value, self.max_value)
        cdf = torch.special.erf((self.support - target[..., None]) / self.sigma_times_sqrt_two)
        z = cdf[..., -1] - cdf[..., 0]
        return (cdf[..., 1:] - cdf[..., :-1]) / z.clamp(min = eps)[..., None]
