"""The distributions used on the hot path (reference src/genjax/distributions.py:22-25, 51-54,
62-65, 92-95, 159-162, 194-197, 298-301).  Each is a `Distribution` whose sampler draws a whole
column from one Threefry key and whose logpdf is a HIP kernel; the formulas are TFP's
(SURVEY.md section 8a, row a3)."""
from __future__ import annotations

import torch

from . import backend as B
from .core import Distribution, as_col
from ._native import GjxError


def _shape_n(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


# normal(mu, sigma)                                            distributions.py:159-162
normal = Distribution(
    lambda key, shape, mu, sigma: B.sample_normal(key, shape, mu, sigma),
    lambda x, mu, sigma: B.normal_logpdf(x, as_col(mu), as_col(sigma)),
    name="normal",
)

# uniform(low, high)                                           distributions.py:194-197
uniform = Distribution(
    lambda key, shape, lo, hi: B.sample_uniform(key, shape, lo, hi),
    lambda x, lo, hi: B.uniform_logpdf(x, as_col(lo), as_col(hi)),
    name="uniform",
)

# flip(p) = Bernoulli(probs=p) (bool in the reference; 0.0/1.0 columns here)   distributions.py:51-54
flip = Distribution(
    lambda key, shape, p: B.sample_bernoulli(key, shape, p),
    lambda x, p: B.bernoulli_logpdf(x.to(torch.float32), as_col(p)),
    name="flip",
)


# bernoulli(logits) = Bernoulli(logits=...)                    distributions.py:22-25
def _bern_logits_sample(key, shape, logits):
    return B.sample_bernoulli(key, shape, torch.sigmoid(as_col(logits)))


def _bern_logits_logpdf(x, logits):
    return B.bernoulli_logpdf(x.to(torch.float32), torch.sigmoid(as_col(logits)))


bernoulli = Distribution(_bern_logits_sample, _bern_logits_logpdf, name="bernoulli")


# beta(a, b)                                                   distributions.py:62-65
def _beta_sample(key, shape, a, b):
    raise GjxError("beta.sample: TFP draws Beta through a data-dependent gamma rejection sampler whose "
                   "stream cannot be reproduced bit for bit (SURVEY.md Appendix C item 4); constrain the "
                   "site or propose it from another distribution")


beta = Distribution(_beta_sample, lambda x, a, b: B.beta_logpdf(x, as_col(a), as_col(b)), name="beta")


# categorical(logits)                                          distributions.py:92-95
class _Categorical(Distribution):
    def _param_event_ndims(self, a):
        return 1

    def sample(self, logits, sample_shape=None, **kwargs):
        from .core import _next_key, batch_size

        logits = as_col(logits)
        key = _next_key()
        if logits.dim() == 2:
            return B.sample_categorical(key, logits)
        n = sample_shape[0] if sample_shape else (batch_size() or 1)
        out = B.sample_categorical(key, logits, n=n)
        return out if (sample_shape or batch_size()) else out[0]


categorical = _Categorical(None, lambda x, logits: B.categorical_logpdf(x, as_col(logits)), name="categorical")


# multivariate_normal(mu, cov)                                 distributions.py:298-301
class _MvNormal(Distribution):
    def sample(self, mu, cov, **kwargs):
        from .core import _next_key, batch_size

        mu, cov = as_col(mu), as_col(cov)
        n = mu.shape[0] if mu.dim() == 2 else (cov.shape[0] if cov.dim() == 3 else (batch_size() or 1))
        out = B.sample_mvnormal(_next_key(), n, mu, cov)
        return out if (mu.dim() == 2 or cov.dim() == 3 or batch_size()) else out[0]

    def logpdf(self, x, mu, cov, **kwargs):
        x = as_col(x)
        single = x.dim() == 1
        out = B.mvnormal_logpdf(x.reshape(-1, x.shape[-1]), as_col(mu), as_col(cov))
        return out[0] if single else out


multivariate_normal = _MvNormal(None, None, name="multivariate_normal")
