"""The distributions of the path at the reference's own test points
(/root/reference/tests/test_distributions.py:126-158, 193-206, 319-370, 558-575, 600-720): assess against
SciPy (standing in for TFP, which is not installed here) at the reference's parameters and samples, finite
draws from simulate, use inside @gen functions, and repeat(n) vectorisation.  Tolerance 1e-5 (reference: 1e-4)."""
import math

import numpy as np
import pytest
import scipy.stats as st
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def G():
    import genjax_amd

    assert torch.cuda.is_available()
    return genjax_amd


def _dev(v, dtype=torch.float32):
    return torch.tensor(v, dtype=dtype, device="cuda")


def _consistent(G, dist, params, samples, expected, seed_key=0):
    for x, want in zip(samples, expected):
        lp, _ = dist.assess(x, *params)
        np.testing.assert_allclose(float(lp), want, atol=TOL, rtol=TOL)
    lp, _ = dist.assess(_dev(samples) if not isinstance(samples, torch.Tensor) else samples, *params)     # a column at once
    np.testing.assert_allclose(lp.cpu().numpy(), np.asarray(expected), atol=TOL, rtol=TOL)
    draw = G.seed(dist.simulate)(G.key(seed_key), *params).get_retval()
    assert bool(torch.isfinite(draw.float()).all())
    return draw


def test_bernoulli_and_flip_consistency(G):
    from genjax_amd import bernoulli, flip

    samples = [0.0, 1.0, 0.0, 1.0]
    p = 1 / (1 + math.exp(-0.5))
    _consistent(G, bernoulli, (0.5,), samples, [math.log(p) if s else math.log(1 - p) for s in samples])
    draw = G.seed(flip.simulate)(G.key(1), 0.7).get_retval()
    assert float(draw) in (0.0, 1.0)       # a 0.0 / 1.0 column (the reference's flip is bool; columns here are 4-byte)
    lp, _ = flip.assess(_dev([True, False, True, False], torch.bool), 0.7)
    np.testing.assert_allclose(lp.cpu().numpy(), [math.log(0.7), math.log(0.3)] * 2, atol=TOL)


def test_categorical_consistency(G):
    from genjax_amd import categorical

    logits = np.array([0.1, 0.6, 0.3])
    logp = logits - np.log(np.exp(logits).sum())
    samples = [0, 1, 2, 1]
    draw = _consistent(G, categorical, (_dev(logits),), _dev(samples, torch.int32), [logp[s] for s in samples])
    assert 0 <= int(draw) <= 2


def test_normal_beta_uniform_consistency(G):
    from genjax_amd import beta, normal, uniform

    xs = [0.5, 1.5, -0.5, 3.0]
    _consistent(G, normal, (1.0, 2.0), xs, st.norm(1.0, 2.0).logpdf(xs))
    xs = [0.2, 0.5, 0.8, 0.1]
    draw = _consistent(G, beta, (2.0, 3.0), xs, st.beta(2.0, 3.0).logpdf(xs))
    assert 0.0 < float(draw) < 1.0
    xs = [0.0, 1.5, -0.5, 2.8]
    draw = _consistent(G, uniform, (-1.0, 3.0), xs, st.uniform(-1.0, 4.0).logpdf(xs))
    assert -1.0 <= float(draw) < 3.0
    lp, _ = uniform.assess(_dev([3.5, -1.5]), -1.0, 3.0)          # outside the support
    assert bool(torch.isinf(lp).all()) and bool((lp < 0).all())


def test_multivariate_normal_consistency(G):
    from genjax_amd import multivariate_normal

    loc, cov = np.array([0.0, 1.0]), np.array([[1.0, 0.5], [0.5, 2.0]])
    xs = np.array([[0.5, 1.5], [-0.5, 0.8], [1.0, 2.0]])
    want = st.multivariate_normal(loc, cov).logpdf(xs)
    for x, w in zip(xs, want):
        lp, _ = multivariate_normal.assess(_dev(x), _dev(loc), _dev(cov))
        np.testing.assert_allclose(float(lp), w, atol=TOL, rtol=TOL)
    lp, _ = multivariate_normal.assess(_dev(xs), _dev(loc), _dev(cov))
    np.testing.assert_allclose(lp.cpu().numpy(), want, atol=TOL, rtol=TOL)
    draw = G.seed(multivariate_normal.simulate)(G.key(2), _dev(loc), _dev(cov)).get_retval()
    assert draw.shape == (2,) and bool(torch.isfinite(draw).all())


def test_distributions_in_generative_functions(G):
    """test_distributions.py:600-630 with this surface's distributions: simulate, then assess of the drawn
    choices returns the sum of the site log-densities."""
    from genjax_amd import beta, categorical, flip, gen, normal, uniform

    @gen
    def model(mu, sigma, a, b):
        x1 = normal(mu, sigma) @ "normal"
        x2 = uniform(0.0, 2.0) @ "uniform"
        x3 = categorical(torch.log(_dev([0.2, 0.3, 0.5]))) @ "categorical"
        x4 = beta(a, b) @ "beta"
        x5 = flip(x4) @ "flip"
        return x1 + x2, x3, x4, x5

    tr = G.seed(model.simulate)(G.key(3), 0.0, 1.0, 2.0, 3.0)
    ch = tr.get_choices()
    assert set(ch) == {"normal", "uniform", "categorical", "beta", "flip"}
    lp, retval = model.assess(ch, 0.0, 1.0, 2.0, 3.0)
    b_ = float(ch["beta"])
    want = (st.norm(0, 1).logpdf(float(ch["normal"])) + math.log(0.5) + math.log([0.2, 0.3, 0.5][int(ch["categorical"])])
            + st.beta(2, 3).logpdf(b_) + math.log(b_ if bool(ch["flip"]) else 1 - b_))
    np.testing.assert_allclose(float(lp), want, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(float(-tr.get_score()), want, rtol=1e-5, atol=1e-5)
    assert float(retval[0]) == float(tr.get_retval()[0])


def test_distributions_vectorization_with_repeat(G):
    """test_distributions.py:692-715: normal.repeat(n) inside a @gen function gives n independent draws under
    one address; the score is the sum of their log-densities; under a particle batch the draws are (N, n)."""
    from genjax_amd import batched, gen, normal

    @gen
    def vector_model(n, mu, sigma):
        return normal.repeat(n)(mu, sigma) @ "samples"

    tr = G.seed(vector_model.simulate)(G.key(4), 100, 0.0, 1.0)
    x = tr.get_retval()
    assert x.shape == (100,) and bool(torch.isfinite(x).all())
    assert tr.get_choices()["samples"].shape == (100,)
    assert len(set(x.cpu().numpy().tolist())) == 100                       # independent draws, not one value repeated
    np.testing.assert_allclose(float(-tr.get_score()), st.norm(0, 1).logpdf(x.cpu().numpy().astype(np.float64)).sum(),
                               rtol=1e-5)
    lp, _ = vector_model.assess({"samples": x}, 100, 0.0, 1.0)
    np.testing.assert_allclose(float(lp), float(-tr.get_score()), rtol=1e-6)
    big = G.seed(vector_model.simulate)(G.key(5), 20000, 1.0, 2.0).get_retval()
    assert abs(float(big.mean()) - 1.0) < 0.05 and abs(float(big.std()) - 2.0) < 0.05
    with batched(7):
        trb = G.seed(vector_model.simulate)(G.key(6), 5, 0.0, 1.0)
    xb = trb.get_retval()
    assert xb.shape == (7, 5) and trb.get_score().shape == (7,)
    np.testing.assert_allclose(-trb.get_score().cpu().numpy(),
                               st.norm(0, 1).logpdf(xb.cpu().numpy().astype(np.float64)).sum(axis=1), rtol=1e-5)
