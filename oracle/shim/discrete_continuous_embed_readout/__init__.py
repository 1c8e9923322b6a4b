"""Stand-in for `discrete-continuous-embed-readout>=0.2.10` (pyproject.toml:31).
MultiCategorical = independent categoricals, one per discrete action type:
Gumbel-max sampling (argmax(logits / T + G), G = -log(-log U)), log-softmax
gather for log_prob, -sum p log p entropy, sum p (log p - log q) KL.  Call
sites: dreamer4.py:1375-1376, 1422-1426, 1478-1481.  The uniform draw goes
through `uniform_like` so the golden generator can inject it.  PARITY UNPINNED
against the real package (its RNG draw order in particular).
Readout: the continuous half only, as ActionEmbedder builds it (dim=None: no projection of its own,
dreamer4.py:1189-1196) — sample / log-prob / entropy of the BetaDist stand-in per continuous action."""
import torch
from torch import nn

from discrete_continuous_embed_readout import discrete_continuous_embed_readout as _impl
from discrete_continuous_embed_readout.discrete_continuous_embed_readout import BetaDist, rescale

def uniform_like(t):
    return torch.rand_like(t)

def _log(t, eps = 1e-20):
    return t.clamp(min = eps).log()

class MultiCategorical:
    def __init__(self, logits, use_parallel_multi_discrete = None):
        self.logits = tuple(logits) if isinstance(logits, (tuple, list)) else (logits,)

    def sample(self, temperature = 1., eps = 1e-10):
        out = []
        for l in self.logits:
            g = -_log(-_log(uniform_like(l)))
            out.append((l / max(temperature, eps) + g).argmax(dim = -1))
        return torch.stack(out, dim = -1)

    def log_prob(self, targets):
        out = []
        for i, l in enumerate(self.logits):
            lp = l.log_softmax(dim = -1)
            t = targets[..., i]
            t = t.expand(lp.shape[:-1]) if t.shape != lp.shape[:-1] else t
            # negative targets are the reference's padding sentinel (D4:7540): every such position is masked out by the caller afterwards,
            # so any in-range index is equivalent there (ASSUMED: the published package does not raise on them)
            out.append(lp.gather(-1, t.clamp(min = 0)[..., None]).squeeze(-1))
        return torch.stack(out, dim = -1)

    def entropy(self):
        out = []
        for l in self.logits:
            lp = l.log_softmax(dim = -1)
            out.append(-(lp.exp() * lp).sum(dim = -1))
        return torch.stack(out, dim = -1)

    def kl_div(self, other, keep_num_actions_dim = False):
        out = []
        for l, m in zip(self.logits, other.logits):
            lp, lq = l.log_softmax(dim = -1), m.log_softmax(dim = -1)
            out.append((lp.exp() * (lp - lq)).sum(dim = -1))
        kl = torch.stack(out, dim = -1)
        return kl if keep_num_actions_dim else kl.sum(dim = -1)

class Readout(nn.Module):
    def __init__(self, dim = None, *, num_discrete = 0, num_continuous = 0, continuous_mean_std = None, continuous_dist_type = 'gaussian',
                 continuous_dist_kwargs: dict = dict(), continuous_squashed = False, **kwargs):
        super().__init__()
        if dim is not None or num_discrete or continuous_dist_type != 'beta' or continuous_squashed or continuous_mean_std is not None:
            raise NotImplementedError('only the projection-free Beta readout of ActionEmbedder is restated')
        self.num_continuous = num_continuous
        self.continuous_dist = BetaDist(**continuous_dist_kwargs)
        self.native_range = (0., 1.)

    def get_selector(self):
        return None                                    # all continuous actions, in order

    def sample_continuous(self, params, selector = None, temperature = 1.):
        return self.continuous_dist.sample(params, temperature = temperature, noise = _impl.beta_noise_like(params.shape[:-1], device = params.device))

    def log_prob_continuous(self, params, targets, selector = None):
        return self.continuous_dist.log_prob(params, targets)

    def entropy_continuous(self, params, selector = None):
        return self.continuous_dist.entropy(params)

    def rescale_from_native(self, actions, target_range):
        return rescale(actions, self.native_range, target_range)
