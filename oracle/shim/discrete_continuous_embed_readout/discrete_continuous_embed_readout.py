"""Stand-in for `discrete_continuous_embed_readout.discrete_continuous_embed_readout` (BetaDist, rescale).
PARITY UNPINNED against the real package (absent from the image, pip is offline): written from the published
description of the package — a Beta policy head whose two raw outputs per action go through softplus, `unimodal=True`
adding 1 so that alpha, beta >= 1 — and from the reference's call sites (dreamer4.py:1172-1196, 1379-1389, 1442-1452,
1491-1494, 4923, 5737).  What is ASSUMED, in one place each:

  * parameterisation   alpha = link(raw[..., 0]) + (1 if unimodal else eps), beta likewise from raw[..., 1]; the link is the module
                       switch BETA_PARAM — 'softplus_p1' (softplus, default) or 'exp_p1' (exp) — read when a BetaDist is built, the
                       descriptor the restatement (Config.continuous_beta_param) and the kernels (d4_config.continuous_beta_param) share
  * sampling           a draw of Beta(alpha_T, beta_T), alpha_T = 1 + (alpha - 1) / T (the density raised to 1 / T and
                       renormalised), realised as Ga / (Ga + Gb) with Marsaglia-Tsang gammas — see `sample` for why
  * native range       (0, 1); `rescale` is the affine map between two ranges

`sample` consumes an explicit noise tensor (normal, uniform) per rejection round so that the reference run, the CPU
restatement and the HIP kernel can be driven with the same draws (torch's own Beta sampler is not injectable)."""
import torch
import torch.nn.functional as F
from torch.distributions import Beta

BETA_PARAM = 'softplus_p1'      # link of the raw parameters; set by oracle/ref_harness.build_reference_model around construction
GAMMA_ROUNDS = 6      # rejection rounds provided per gamma draw (acceptance >= 0.95 per round for shape >= 1)


def rescale(t, from_range, to_range):
    (a, b), (c, d) = from_range, to_range
    return (t - a) / (b - a) * (d - c) + c


def beta_noise_like(shape, device=None):
    """Noise of one `sample` call: (*shape, 2 gammas, GAMMA_ROUNDS, 2) with [..., 0] standard normal and [..., 1] uniform(0, 1).
    The golden generator replaces this function to inject the draws."""
    n = torch.randn(*shape, 2, GAMMA_ROUNDS, device=device)
    u = torch.rand(*shape, 2, GAMMA_ROUNDS, device=device)
    return torch.stack((n, u), dim=-1)


def gamma_from_noise(shape_param, noise):
    """Marsaglia & Tsang (2000) for shape >= 1 with the rejection loop unrolled over the provided rounds: the first accepted
    candidate wins; if every round rejects (probability < 1e-7) the last candidate is taken.  noise (..., rounds, 2)."""
    d = shape_param - 1. / 3.
    c = 1. / torch.sqrt(9. * d)
    out = None
    done = torch.zeros_like(shape_param, dtype=torch.bool)
    rounds = noise.shape[-2]
    for r in range(rounds):
        x, u = noise[..., r, 0], noise[..., r, 1]
        t = 1. + c * x
        v = t * t * t
        ok = (v > 0.) & (torch.log(u.clamp(min=1e-30)) < 0.5 * x * x + d - d * v + d * torch.log(v.clamp(min=1e-30)))
        cand = d * v
        take = (ok | (r == rounds - 1)) & ~done
        out = cand if out is None else torch.where(take, cand, out)
        done = done | ok
    return out.clamp(min=1e-30)


class BetaDist:
    def __init__(self, unimodal=False, eps=1e-6):
        self.unimodal, self.eps = unimodal, eps
        assert BETA_PARAM in ('softplus_p1', 'exp_p1')
        self.link = torch.exp if BETA_PARAM == 'exp_p1' else F.softplus

    def alpha_beta(self, params):
        base = 1. if self.unimodal else self.eps
        return self.link(params[..., 0]) + base, self.link(params[..., 1]) + base

    def dist(self, params):
        return Beta(*self.alpha_beta(params))

    def sample(self, params, temperature=1., noise=None):
        a, b = self.alpha_beta(params)
        assert self.unimodal, 'the injectable sampler needs alpha, beta >= 1'
        t = max(float(temperature), 1e-10)
        a, b = 1. + (a - 1.) / t, 1. + (b - 1.) / t
        if noise is None:
            noise = beta_noise_like(a.shape, device=a.device)
        ga, gb = gamma_from_noise(a, noise[..., 0, :, :]), gamma_from_noise(b, noise[..., 1, :, :])
        return ga / (ga + gb)

    def log_prob(self, params, value):
        return self.dist(params).log_prob(value)

    def entropy(self, params):
        return self.dist(params).entropy()
