"""The distributions used on the hot path (reference src/genjax/distributions.py:22-25, 51-54,
62-65, 92-95, 159-162, 194-197, 298-301).  Each is a `Distribution` whose sampler draws a whole
column from one Threefry key and whose logpdf is a HIP kernel; the formulas are TFP's
(SURVEY.md section 8a, row a3)."""
from __future__ import annotations

import torch

from . import backend as B
from .core import Distribution, as_col


def _shape_n(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


# --------------------------------------------------------------------------- #
# log-densities that autograd can differentiate: generic hmc (mcmc.py:434-578) takes jax.grad of assess();
# here the HIP log-density kernel is the forward and the HIP gradient kernel (gjx_logpdf_grad_f32) the
# backward of a torch.autograd.Function, so that the user's own glue between the sites (a + b * x, ...) is
# differentiated by autograd and the densities by the library.
# --------------------------------------------------------------------------- #
_FORWARD = {"normal": B.normal_logpdf, "uniform": B.uniform_logpdf, "beta": B.beta_logpdf,
            "bernoulli": lambda x, p, _unused: B.bernoulli_logpdf(x, p)}


class _LogpdfFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, x, p1, p2):
        ctx.kind = kind
        ctx.save_for_backward(x, p1, p2)
        return _FORWARD[kind](x, p1, p2)

    @staticmethod
    def backward(ctx, g):
        x, p1, p2 = ctx.saved_tensors
        want = tuple(ctx.needs_input_grad[1:4])
        if ctx.kind == "bernoulli":
            want = (False, want[1], False)
        gx, g1, g2 = B.logpdf_grad(ctx.kind, x, p1, None if ctx.kind == "bernoulli" else p2, g.contiguous(), want=want)
        fit = lambda gr, like: None if gr is None else gr.sum_to_size(like.shape)   # undo the broadcast
        return None, fit(gx, x), fit(g1, p1), fit(g2, p2)


def _logpdf(kind, x, p1, p2=None):
    x, p1 = as_col(x), as_col(p1)
    p2 = as_col(p2) if p2 is not None else torch.zeros((), dtype=torch.float32, device=x.device)
    if torch.is_grad_enabled() and any(t.requires_grad for t in (x, p1, p2)):
        return _LogpdfFn.apply(kind, x, p1, p2)
    return _FORWARD[kind](x, p1, p2)


# normal(mu, sigma)                                            distributions.py:159-162
normal = Distribution(
    lambda key, shape, mu, sigma: B.sample_normal(key, shape, mu, sigma),
    lambda x, mu, sigma: _logpdf("normal", x, mu, sigma),
    name="normal",
)

# uniform(low, high)                                           distributions.py:194-197
uniform = Distribution(
    lambda key, shape, lo, hi: B.sample_uniform(key, shape, lo, hi),
    lambda x, lo, hi: _logpdf("uniform", x, lo, hi),
    name="uniform",
)

# flip(p) = Bernoulli(probs=p) (bool in the reference; 0.0/1.0 columns here)   distributions.py:51-54
flip = Distribution(
    lambda key, shape, p: B.sample_bernoulli(key, shape, p),
    lambda x, p: _logpdf("bernoulli", x.to(torch.float32), p),
    name="flip",
)


# bernoulli(logits) = Bernoulli(logits=...)                    distributions.py:22-25
# log_prob = -x softplus(-l) - (1 - x) softplus(l) ([3P] TFP), finite for any |l|; sample = uniform < sigmoid(l)
bernoulli = Distribution(
    lambda key, shape, logits: B.sample_bernoulli_logits(key, shape, logits),
    lambda x, logits: B.bernoulli_logits_logpdf(x.to(torch.float32), as_col(logits)),
    name="bernoulli",
)


# beta(a, b)                                                   distributions.py:62-65
# [3P] tfd.Beta samples through a data-dependent gamma rejection sampler whose stream is not reproduced
# (SURVEY.md Appendix C item 4): the draw has the same law (Marsaglia-Tsang gammas on counter-based draws of
# the site key), not the same bits.
beta = Distribution(
    lambda key, shape, a, b: B.sample_beta(key, shape, a, b),
    lambda x, a, b: _logpdf("beta", x, a, b),
    name="beta",
)


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
        x, mu, cov = as_col(x), as_col(mu), as_col(cov)
        single = x.dim() == 1 and mu.dim() == 1 and cov.dim() == 2
        out = B.mvnormal_logpdf(x.reshape(-1, x.shape[-1]), mu, cov)
        return out[0] if single else out


multivariate_normal = _MvNormal(None, None, name="multivariate_normal")
