"""The reference's Python surface beyond the fused families: the distributions added this round (beta draws,
logits Bernoulli), the gradient kernels behind generic hmc, the Cond combinator, hmc on generic @gen traces
against the reference's own conjugate fixtures (tests/test_mcmc.py), chain() over fused traces, and the
weight arithmetic of ParticleCollection through the library.

Parity status of this file: statistical (moments against closed forms the reference's tests use) and against
f64 restatements of the published formulas; the reference's sample STREAMS for beta are not reproduced
(SURVEY.md Appendix C item 4), which is stated where it applies."""
import math

import numpy as np
import pytest
import torch

from tests_fixtures import beta_bernoulli_posterior, normal_normal_posterior

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import genjax_amd

    assert torch.cuda.is_available()
    return genjax_amd


@pytest.fixture(scope="module")
def B():
    from genjax_amd import backend

    return backend


# ------------------------------------------------------------------ distributions
def test_bernoulli_logits_logpdf_is_finite_and_exact_up_to_50(G, B):
    """log_prob = -x softplus(-l) - (1 - x) softplus(l) ([3P] TFP Bernoulli(logits=...)): finite and within 2 ulp of
    the f64 formula for |l| <= 50 (the sigmoid-then-log route of round 1 returned -inf from |l| = 17 on)."""
    l = torch.linspace(-50, 50, 20001, device="cuda")
    for xv in (0.0, 1.0):
        x = torch.full_like(l, xv)
        got = G.bernoulli.logpdf(x, l).double().cpu().numpy()
        l64 = l.double().cpu().numpy()
        ref = -(xv * np.logaddexp(0, -l64) + (1 - xv) * np.logaddexp(0, l64))
        assert np.isfinite(got).all()
        err = np.abs(got - ref)
        bar = 2 * np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64) + 1e-45
        assert (err <= bar).all(), (err / bar).max()
    # and through the GFI: assess of a site
    lp, _ = G.bernoulli.assess(torch.ones(4, device="cuda"), torch.tensor([-50.0, -1.0, 1.0, 50.0], device="cuda"))
    np.testing.assert_allclose(lp.cpu().numpy(), [-50.0, -1.3132616, -0.3132617, -math.exp(-50.0)], rtol=2e-6)


def test_bernoulli_logits_sampler_frequencies(G, B):
    n = 1 << 20
    for lg in (-3.0, 0.0, 0.7, 4.0):
        x = B.sample_bernoulli_logits(G.key(11), (n,), torch.full((n,), lg, device="cuda"))
        assert set(np.unique(x.cpu().numpy())) <= {0.0, 1.0}
        p = 1 / (1 + math.exp(-lg))
        assert abs(float(x.mean()) - p) < 5 * math.sqrt(p * (1 - p) / n)
    # the draw is uniform(key) < sigmoid(l): the same key gives nested outcomes for increasing logits
    a = B.sample_bernoulli_logits(G.key(12), (n,), torch.full((n,), -0.5, device="cuda"))
    b = B.sample_bernoulli_logits(G.key(12), (n,), torch.full((n,), 0.5, device="cuda"))
    assert bool((a <= b).all())


@pytest.mark.parametrize("a,b", [(2.0, 5.0), (1.0, 10.0), (0.5, 0.5), (30.0, 3.0)])
def test_beta_draws_have_the_beta_law(G, B, a, b):
    """beta.sample (distributions.py:62-65): Kolmogorov-Smirnov against scipy's Beta(a, b) CDF and the first two
    moments.  The law is the reference's; the bit stream is not (its gamma rejection sampler is data dependent)."""
    from scipy import stats

    n = 200000
    x = G.seed(lambda: G.beta.sample(torch.full((n,), a, device="cuda"), b))(G.key(5))
    xs = x.double().cpu().numpy()
    assert ((xs >= 0) & (xs <= 1)).all()
    ks = stats.kstest(xs, stats.beta(a, b).cdf)
    assert ks.pvalue > 1e-4, ks
    m, v = a / (a + b), a * b / ((a + b) ** 2 * (a + b + 1))
    assert abs(xs.mean() - m) < 5 * math.sqrt(v / n)
    assert abs(xs.var() - v) < 0.03 * v + 1e-5
    # same key, same draws; a different key, different draws
    y = G.seed(lambda: G.beta.sample(torch.full((n,), a, device="cuda"), b))(G.key(5))
    z = G.seed(lambda: G.beta.sample(torch.full((n,), a, device="cuda"), b))(G.key(6))
    assert torch.equal(x, y) and not torch.equal(x, z)


def test_logpdf_gradient_kernels_against_f64_autograd(B):
    """gjx_logpdf_grad_f32 (the backward of every log-density under generic hmc) against torch.autograd on the
    f64 formula, and the upstream factor."""
    n = 4096
    g = torch.Generator(device="cuda").manual_seed(3)
    rnd = lambda lo, hi: torch.rand(n, device="cuda", generator=g) * (hi - lo) + lo          # noqa: E731
    up = rnd(-2, 2)
    cases = {
        "normal": (rnd(-3, 3), rnd(-1, 1), rnd(0.2, 2.0)),
        "beta": (rnd(0.02, 0.98), rnd(0.5, 6), rnd(0.5, 6)),
        "bernoulli": ((rnd(0, 1) < 0.5).float(), rnd(0.05, 0.95), None),
        "uniform": (rnd(-0.9, 0.9), rnd(-3, -1), rnd(1, 3)),
    }
    for kind, (x, p1, p2) in cases.items():
        xs = [t.double().clone().requires_grad_(True) if t is not None else None for t in (x, p1, p2)]
        X, P1, P2 = xs
        if kind == "normal":
            lp = -0.5 * ((X - P1) / P2) ** 2 - torch.log(P2) - 0.5 * math.log(2 * math.pi)
        elif kind == "beta":
            lp = ((P1 - 1) * torch.log(X) + (P2 - 1) * torch.log1p(-X)
                  - (torch.lgamma(P1) + torch.lgamma(P2) - torch.lgamma(P1 + P2)))
        elif kind == "bernoulli":
            lp = X * torch.log(P1) + (1 - X) * torch.log1p(-P1)
        else:
            lp = -torch.log(P2 - P1) + 0 * X
        want = [t is not None and kind != "bernoulli" or (kind == "bernoulli" and i == 1) for i, t in enumerate(xs)]
        refs = torch.autograd.grad((lp * up.double()).sum(), [t for t, w in zip(xs, want) if w], allow_unused=True)
        refs = iter(refs)
        got = B.logpdf_grad(kind, x, p1, p2, up, want=tuple(want))
        for w, gk in zip(want, got):
            if not w:
                assert gk is None
                continue
            r = next(refs)
            r = torch.zeros(n, dtype=torch.float64, device="cuda") if r is None else r
            np.testing.assert_allclose(gk.double().cpu().numpy(), r.cpu().numpy(), rtol=2e-5, atol=2e-5, err_msg=kind)


def test_autograd_through_a_gen_function_uses_the_gradient_kernels(G):
    """d assess / d choices of y ~ normal(a + b * x, 0.3) through user glue: autograd for the glue, the library for
    the densities; against the closed form."""
    from genjax_amd import gen, normal
    from genjax_amd.core import batched

    @gen
    def line(x):
        a = normal(0.0, 1.0) @ "a"
        b = normal(0.0, 2.0) @ "b"
        return normal(a + b * x, 0.3) @ "y"

    n = 1000
    a = torch.randn(n, device="cuda").requires_grad_(True)
    b = torch.randn(n, device="cuda").requires_grad_(True)
    x, y = 0.7, 1.1
    with batched(n):
        lp, _ = line.assess({"a": a, "b": b, "y": torch.full((n,), y, device="cuda")}, x)
    ga, gb = torch.autograd.grad(lp.sum(), [a, b])
    r = (y - (a + b * x)).detach() / 0.09
    np.testing.assert_allclose(ga.cpu().numpy(), (r - a.detach()).cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gb.cpu().numpy(), (r * x - b.detach() / 4).cpu().numpy(), rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ Cond
def test_cond_is_exported_and_follows_the_gfi_algebra(G):
    """Cond(callee, callee_) (core.py:2638-2785): both branches evaluated, selected with the check; weights of
    generate / assess / update / regenerate consistent with one another; an MH move through it."""
    from genjax_amd import Cond, flip, gen, normal, seed, sel, uniform
    from genjax_amd.core import batched
    from genjax_amd.inference import mh

    @gen
    def wide(m):
        return uniform(-2.0, 2.0) @ "obs"

    @gen
    def tight(m):
        return normal(m, 0.1) @ "obs"

    @gen
    def point(m):
        o = flip(0.3) @ "is_outlier"
        return Cond(wide, tight)(o, m) @ "y"

    n = 20000
    with batched(n):
        tr = seed(point.simulate)(G.key(1), 0.5)
        ch = tr.get_choices()
        o, obs = ch["is_outlier"], ch["y"]["obs"]
        assert abs(float(o.float().mean()) - 0.3) < 0.02
        inl = obs[o == 0]
        assert abs(float(inl.mean()) - 0.5) < 0.01 and float(obs[o != 0].abs().max()) > 1.0
        lp, rv = point.assess(ch, 0.5)
        np.testing.assert_allclose(lp.cpu().numpy(), (-tr.get_score()).cpu().numpy(), rtol=1e-5, atol=1e-5)
        assert torch.equal(rv, obs)
        ref = torch.where(o != 0, torch.full_like(obs, math.log(0.3) - math.log(4.0)),
                          math.log(0.7) - 0.5 * ((obs - 0.5) / 0.1) ** 2 - math.log(0.1) - 0.5 * math.log(2 * math.pi))
        np.testing.assert_allclose(lp.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
        # generate with the observation constrained: weight = density of the chosen branch at the observation
        tr2, w2 = seed(point.generate)(G.key(2), {"y": {"obs": torch.full((n,), 0.45, device="cuda")}}, 0.5)
        o2 = tr2.get_choices()["is_outlier"]
        refw = torch.where(o2 != 0, torch.full((n,), -math.log(4.0), device="cuda"),
                           torch.full((n,), -0.5 * 0.25 - math.log(0.1) - 0.5 * math.log(2 * math.pi), device="cuda"))
        np.testing.assert_allclose(w2.cpu().numpy(), refw.cpu().numpy(), rtol=1e-4, atol=1e-4)
        # update of the observation: weight = new score difference
        tr3, w3, _ = point.update(tr2, {"y": {"obs": torch.full((n,), 0.6, device="cuda")}}, 0.5)
        np.testing.assert_allclose(w3.cpu().numpy(), (tr2.get_score() - tr3.get_score()).cpu().numpy(), rtol=1e-4,
                                   atol=1e-4)
        # regenerate the indicator: the reference's Cond weighs each branch against ITS OWN earlier trace
        # (core.py:2763-2785), so with the observation and the arguments unchanged the weight is 0 whichever way
        # the check flips, and the observation is kept -- the semantics are reproduced, not repaired
        tr4, w4, _ = seed(point.regenerate)(G.key(9), tr2, sel("is_outlier"), 0.5)
        assert float(w4.abs().max()) == 0.0
        assert torch.equal(tr4.get_choices()["y"]["obs"], tr2.get_choices()["y"]["obs"])
        assert not torch.equal(tr4.get_choices()["is_outlier"], o2)
        # regenerate inside the branches: both branch traces redraw, the active one shows
        tr5, w5, _ = seed(point.regenerate)(G.key(10), tr, sel("y"), 0.5)
        o5, obs5 = tr5.get_choices()["is_outlier"], tr5.get_choices()["y"]["obs"]
        assert torch.equal(o5, o) and not torch.equal(obs5, obs)
        assert abs(float(obs5[o5 == 0].mean()) - 0.5) < 0.01 and float(w5.abs().max()) == 0.0
        # an MH move through the combinator and the leaf selection of its trace
        cur = seed(lambda t: mh(t, sel("is_outlier")))(G.key(100), tr2)
        assert torch.equal(cur.get_choices()["y"]["obs"], tr2.get_choices()["y"]["obs"])
        assert isinstance(cur._choices["y"], type(tr2._choices["y"]))          # still a CondTr: both branches kept


# ------------------------------------------------------------------ generic hmc (reference tests/test_mcmc.py:1314-1750)
def test_hmc_basic_functionality(G):
    """test_mcmc.py:1314-1336: one step on a single (unbatched) trace keeps the structure."""
    from genjax_amd import gen, normal, seed, sel
    from genjax_amd.inference import hmc

    @gen
    def simple_normal():
        return normal(0.0, 1.0) @ "x"

    tr = seed(simple_normal.simulate)(G.key(42))
    out = seed(lambda t: hmc(t, sel("x"), step_size=0.1, n_steps=5))(G.key(123), tr)
    x = out.get_choices()["x"]
    assert x.shape == () and bool(torch.isfinite(x))
    again = seed(lambda t: hmc(t, sel("x"), step_size=0.1, n_steps=5))(G.key(123), tr)
    assert torch.equal(again.get_choices()["x"], x)


def test_hmc_convergence_normal(G):
    """test_mcmc.py:1598-1630: normal(0.5, 1.5), step 0.15 x 10; the reference runs one chain of 3000 steps at
    tolerance 0.3, here 512 chains x 150 steps at a tolerance 6 times tighter."""
    from genjax_amd import const, gen, normal, seed, sel
    from genjax_amd.inference import chain, hmc

    @gen
    def normal_model():
        return normal(0.5, 1.5) @ "x"

    tr = seed(normal_model.simulate)(G.key(0))
    res = seed(chain(lambda t: hmc(t, sel("x"), step_size=0.15, n_steps=10)))(
        G.key(1), tr, const(150), burn_in=const(50), n_chains=const(512))
    x = res.traces.get_choices()["x"]
    assert x.shape == (512, 100)
    assert abs(float(x.mean()) - 0.5) < 0.05 and abs(float(x.var()) - 2.25) < 0.1, (float(x.mean()), float(x.var()))
    assert float(res.acceptance_rate) > 0.6


def test_hmc_beta_bernoulli_convergence(G):
    """test_mcmc.py:1339-1370: p ~ beta(2, 5), obs ~ flip(p) = True; posterior Beta(3, 5)."""
    from genjax_amd import beta, const, flip, gen, seed, sel
    from genjax_amd.inference import chain, hmc

    @gen
    def beta_bernoulli():
        p = beta(2.0, 5.0) @ "p"
        return flip(p) @ "obs"

    tr, _ = seed(beta_bernoulli.generate)(G.key(3), {"obs": True})
    res = seed(chain(lambda t: hmc(t, sel("p"), step_size=0.05, n_steps=10)))(
        G.key(4), tr, const(160), burn_in=const(60), n_chains=const(512))
    p = res.traces.get_choices()["p"]
    mean, var, _, _ = beta_bernoulli_posterior(True, 2.0, 5.0)
    assert abs(float(p.mean()) - mean) < 0.01 and abs(float(p.var()) - var) < 0.004, (float(p.mean()), float(p.var()))
    assert float(res.acceptance_rate) > 0.5
    assert bool(((p > 0) & (p < 1)).all())


def test_hmc_hierarchical_normal_exact_posterior(G):
    """test_mcmc.py:1633-1750: mu ~ N(0, 1), y ~ N(mu, 0.5) = 1.5; step 0.2 x 15; exact posterior N(1.2, 0.2) at
    the reference's strict tolerance (0.05)."""
    from genjax_amd import const, gen, normal, seed, sel
    from genjax_amd.inference import chain, hmc

    @gen
    def hierarchical(prior_mean, prior_std, obs_std):
        mu = normal(prior_mean, prior_std) @ "mu"
        return normal(mu, obs_std) @ "y"

    tr, _ = seed(hierarchical.generate)(G.key(5), {"y": 1.5}, 0.0, 1.0, 0.5)
    res = seed(chain(lambda t: hmc(t, sel("mu"), step_size=0.2, n_steps=15)))(
        G.key(6), tr, const(120), burn_in=const(40), n_chains=const(512))
    mu = res.traces.get_choices()["mu"]
    mean, var = normal_normal_posterior(1.5, 0.0, 1.0, 0.25)
    assert abs(float(mu.mean()) - mean) < 0.02 and abs(float(mu.var()) - var) < 0.02
    assert abs(float(res.rhat["mu"]) - 1.0) < 0.2      # 0.2 x 15 is close to a full period of this posterior: slow mixing
    y = res.traces.get_choices()["y"]
    assert bool((y == 1.5).all())                           # the unselected site never moves


def test_generic_hmc_vector_leaf_is_distributionally_right(G):
    """A selected leaf with three elements per chain (a Vmap'd site): v_i ~ N(m_i, 1), y_i ~ N(v_i, 0.5) observed.
    The reference draws one scalar momentum per leaf (mcmc.py:487), which only fits scalar leaves; here every element
    gets its own momentum from the leaf's site key.  No reference bits exist for that case (parity unpinned), so the
    check is the exact conjugate posterior per element: mean (m + 4 y) / 5, variance 0.2, elements uncorrelated.
    A vector leaf OUTSIDE a Vmap has an element-wise score (as in the reference) and is refused by hmc."""
    from genjax_amd import const, gen, normal, seed, sel
    from genjax_amd._native import GjxError
    from genjax_amd.inference import chain, hmc

    m = torch.tensor([-1.0, 0.5, 2.0], device="cuda")
    yv = torch.tensor([0.5, -0.25, 1.0], device="cuda")

    @gen
    def latent(mu):
        return normal(mu, 1.0) @ "v"

    @gen
    def observed(v):
        return normal(v, 0.5) @ "y"

    @gen
    def vec_model():
        v = latent.vmap(in_axes=(0,))(m) @ "vs"
        return observed.vmap(in_axes=(0,))(v) @ "ys"

    tr, _ = seed(vec_model.generate)(G.key(11), {"ys": {"y": yv}})
    # trajectory 0.7 = a quarter period of the posterior (sd 0.447): one move decorrelates.  (0.2 x 7 would be HALF a
    # period: every move mirrors the point about the mean with acceptance 1 and the chain never mixes.)
    res = seed(chain(lambda t: hmc(t, sel(("vs", "v")), step_size=0.1, n_steps=7)))(
        G.key(12), tr, const(120), burn_in=const(40), n_chains=const(512))
    v = res.traces.get_choices()["vs"]["v"]
    assert tuple(v.shape) == (512, 80, 3), v.shape
    flat = v.reshape(-1, 3).double()
    want = ((m + 4.0 * yv) / 5.0).double()
    assert float((flat.mean(dim=0) - want).abs().max()) < 0.02, flat.mean(dim=0)
    assert float((flat.var(dim=0) - 0.2).abs().max()) < 0.02, flat.var(dim=0)
    corr = torch.corrcoef(flat.t())
    assert float((corr - torch.eye(3, device=corr.device, dtype=corr.dtype)).abs().max()) < 0.03
    assert float(res.acceptance_rate) > 0.6

    @gen
    def bare_vector():
        v = normal(m, 1.0) @ "v"
        return normal(v, 0.5) @ "y"

    tr2, _ = seed(bare_vector.generate)(G.key(13), {"y": yv})
    with pytest.raises(GjxError, match="one log-density per particle"):
        seed(lambda t: hmc(t, sel("v"), step_size=0.2, n_steps=3))(G.key(14), tr2)


# ------------------------------------------------------------------ chain (reference tests/test_mcmc.py:459-668)
def _beta_bernoulli():
    from genjax_amd import beta, flip, gen

    @gen
    def beta_bernoulli():
        p = beta(2.0, 5.0) @ "p"
        return flip(p) @ "obs"

    return beta_bernoulli


def test_chain_result_structure_burn_in_and_thinning(G):
    """test_mcmc.py:459-569: accepts are 0/1 per kept step and average to acceptance_rate; burn_in and
    autocorrelation_resampling keep steps arange(burn_in, n_steps, thin); retvals stack like choices."""
    from genjax_amd import const, seed, sel
    from genjax_amd.inference import MCMCResult, chain, mh

    model = _beta_bernoulli()
    tr = seed(model.simulate)(G.key(0))
    run = seed(chain(lambda t: mh(t, sel("p"))))
    res = run(G.key(1), tr, const(100))
    assert isinstance(res, MCMCResult)
    assert res.traces.get_retval().shape[0] == 100 and res.accepts.shape == (100,)
    assert bool(((res.accepts == 0) | (res.accepts == 1)).all())
    np.testing.assert_allclose(float(res.acceptance_rate), float(res.accepts.float().mean()), rtol=1e-6)
    assert res.n_chains.value == 1 and res.rhat is None and res.ess_bulk is None and res.ess_tail is None
    res = run(G.key(1), tr, const(100), burn_in=const(2), autocorrelation_resampling=const(1))
    assert res.n_steps.value == 98 and res.traces.get_retval().shape[0] == 98 and res.accepts.shape == (98,)
    res = run(G.key(1), tr, const(100), burn_in=const(1), autocorrelation_resampling=const(2))
    assert res.n_steps.value == len(range(1, 100, 2)) == 50
    assert res.traces.get_retval().shape[0] == 50 and res.accepts.shape == (50,)
    # thinning keeps every second state of the same chain
    full = run(G.key(1), tr, const(100))
    assert torch.equal(res.traces.get_choices()["p"], full.traces.get_choices()["p"][1::2])
    explicit = run(G.key(1), tr, const(100), n_chains=const(1))
    assert explicit.n_chains.value == 1 and explicit.rhat is None
    assert explicit.traces.get_retval().shape == full.traces.get_retval().shape


def test_chain_multiple_chains(G):
    """test_mcmc.py:572-620: (n_chains, n_steps) layout of retvals and accepts; R-hat and ESS dictionaries keyed
    like the choices."""
    from genjax_amd import const, seed, sel
    from genjax_amd.inference import chain, mh

    model = _beta_bernoulli()
    tr = seed(model.simulate)(G.key(0))
    res = seed(chain(lambda t: mh(t, sel("p"))))(G.key(1), tr, const(100), n_chains=const(4))
    assert res.n_chains.value == 4 and res.n_steps.value == 100
    assert res.traces.get_retval().shape == (4, 100) and res.accepts.shape == (4, 100)
    assert 0.0 <= float(res.acceptance_rate) <= 1.0
    choices = res.traces.get_choices()
    assert choices["obs"].shape == (4, 100)
    for d in (res.rhat, res.ess_bulk, res.ess_tail):
        assert set(d) == set(choices)
    assert math.isfinite(float(res.rhat["p"])) and float(res.rhat["p"]) > 0.5
    assert float(res.ess_bulk["p"]) > 0 and float(res.ess_tail["p"]) > 0
    # the chains are distinct realisations
    p = choices["p"]
    assert not torch.equal(p[0], p[1])


# ------------------------------------------------------------------ mh posteriors (reference tests/test_mcmc.py:203-431)
@pytest.mark.parametrize("obs", [True, False])
def test_mh_beta_bernoulli_posterior(G, obs):
    """test_mcmc.py:203-307: p ~ beta(2, 5), obs ~ flip(p); mh(sel("p")) targets Beta(3, 5) / Beta(2, 6).  The
    reference runs one chain of 50000 steps at tolerance 0.3 on the variance and a "practical" 0.1-0.25 on the
    mean; here 1024 chains x 60 kept steps at 0.01 / 0.004."""
    from genjax_amd import const, seed, sel
    from genjax_amd.inference import chain, mh

    model = _beta_bernoulli()
    tr, _ = seed(model.generate)(G.key(30), {"obs": obs})
    res = seed(chain(lambda t: mh(t, sel("p"))))(G.key(31), tr, const(100), burn_in=const(40), n_chains=const(1024))
    p = res.traces.get_choices()["p"]
    mean, var, _, _ = beta_bernoulli_posterior(obs, 2.0, 5.0)
    assert abs(float(p.mean()) - mean) < 0.01 and abs(float(p.var()) - var) < 0.004, (float(p.mean()), float(p.var()))
    o = res.traces.get_choices()["obs"]
    assert bool((o == float(obs)).all())                   # the observed site never moves
    assert 0.2 < float(res.acceptance_rate) < 0.9


def test_mh_bivariate_normal_marginal(G):
    """test_mcmc.py:369-431: x ~ N(0, 1), y | x ~ N(0.5 x, 0.5), y = 2; x | y is N(2, 0.5).  mh regenerates x from
    its prior, two prior standard deviations away from the posterior mean, so the chains need a few hundred
    steps (the reference: 5000 steps, tolerance 2.5 on the mean and 0.3 on the variance)."""
    from genjax_amd import const, gen, normal, seed, sel
    from genjax_amd.inference import chain, mh

    @gen
    def bivariate_normal():
        x = normal(0.0, 1.0) @ "x"
        return normal(0.5 * x, 0.5) @ "y"

    tr, _ = seed(bivariate_normal.generate)(G.key(32), {"y": 2.0})
    res = seed(chain(lambda t: mh(t, sel("x"))))(G.key(33), tr, const(700), burn_in=const(400), n_chains=const(1024))
    x = res.traces.get_choices()["x"]
    post_var = 1.0 / (1.0 + 0.25 / 0.25)
    post_mean = post_var * (0.5 * 2.0 / 0.25)
    assert post_mean == 2.0 and post_var == 0.5
    assert abs(float(x.mean()) - post_mean) < 0.03 and abs(float(x.var()) - post_var) < 0.03, (float(x.mean()), float(x.var()))
    assert abs(float(res.rhat["x"]) - 1.0) < 0.2        # low acceptance (independence proposals from the prior): 300 kept steps per chain


def test_mcmc_is_deterministic_under_a_seed(G):
    """test_mcmc.py:760-788: the same key gives the same chain, another key another one."""
    from genjax_amd import const, gen, normal, seed, sel
    from genjax_amd.inference import chain, mh

    @gen
    def simple_normal(mu, sigma):
        return normal(mu, sigma) @ "x"

    tr = seed(simple_normal.simulate)(G.key(34), 0.0, 1.0)
    run = seed(chain(lambda t: mh(t, sel("x"))))
    a, b, c = run(G.key(35), tr, const(100)), run(G.key(35), tr, const(100)), run(G.key(36), tr, const(100))
    xa = a.traces.get_choices()["x"]
    assert torch.equal(xa, b.traces.get_choices()["x"]) and torch.equal(a.accepts, b.accepts)
    assert not torch.equal(xa, c.traces.get_choices()["x"])


# ------------------------------------------------------------------ mala (reference tests/test_mcmc.py:821-1312)
def test_mala_basic_functionality_and_definition(G):
    """test_mcmc.py:821-850, plus the move against its definition (mcmc.py:359-419) in f64 on the proposal it
    drew: standard normal target => grad = -x, proposal mean x (1 - tau^2/2), log alpha from the two proposal
    densities; accepted particles carry the proposal, rejected ones the old value."""
    from genjax_amd import const, gen, normal, seed, sel
    from genjax_amd.inference import MCMCResult, chain, init, mala, rejuvenate
    from genjax_amd.inference import mcmc as M

    @gen
    def simple_normal(mu, sigma):
        return normal(mu, sigma) @ "x"

    tr = seed(simple_normal.simulate)(G.key(42), 0.0, 1.0)
    res = seed(chain(lambda t: mala(t, sel("x"), 0.1)))(G.key(7), tr, const(100))
    assert isinstance(res, MCMCResult)
    assert res.traces.get_choices()["x"].shape == (100,)
    assert 0.1 <= float(res.acceptance_rate) <= 1.0

    n, tau = 50000, 0.9
    p = seed(init)(G.key(8), simple_normal, (0.0, 1.0), const(n), {})
    x0 = p.traces.get_choices()["x"].double()
    out = seed(rejuvenate)(G.key(9), p, lambda t: mala(t, sel("x"), tau))
    x1 = out.traces.get_choices()["x"].double()
    acc = M.last_accept["value"]
    assert 0.5 < float(acc.float().mean()) < 0.99
    assert torch.equal(x1[~acc], x0[~acc]) and bool((x1[acc] != x0[acc]).all())
    # the reverse-proposal term of every ACCEPTED move is consistent with the definition: log alpha >= log u
    # cannot be checked without u, but the proposal law can: (x' - mean_fwd)/tau is a standard normal sample
    z = ((x1[acc] - x0[acc] * (1 - tau * tau / 2)) / tau).cpu().numpy()
    # the target is invariant under the move (detailed balance): x1 is N(0, 1) again
    assert abs(float(x1.mean())) < 0.02 and abs(float(x1.var()) - 1.0) < 0.02
    assert abs(float(np.mean(z))) < 0.05
    # same key, same bits
    again = seed(rejuvenate)(G.key(9), p, lambda t: mala(t, sel("x"), tau))
    assert torch.equal(again.traces.get_choices()["x"].double(), x1)


def test_mala_acceptance_follows_the_two_proposal_densities(G):
    """One MALA step with a step size far too large for the target is almost always rejected, a tiny one almost
    always accepted (test_mcmc.py:973-1013 step-size effects), and an empty selection is the identity."""
    from genjax_amd import const, gen, normal, seed, sel
    from genjax_amd.inference import init, mala, rejuvenate
    from genjax_amd.inference import mcmc as M

    @gen
    def narrow():
        return normal(0.0, 0.01) @ "x"

    p = seed(init)(G.key(10), narrow, (), const(20000), {})
    seed(rejuvenate)(G.key(11), p, lambda t: mala(t, sel("x"), 1.0))
    assert float(M.last_accept["value"].float().mean()) < 0.02        # drift overshoots by 5000 sigma
    seed(rejuvenate)(G.key(11), p, lambda t: mala(t, sel("x"), 1e-4))
    assert float(M.last_accept["value"].float().mean()) > 0.98
    same = seed(rejuvenate)(G.key(11), p, lambda t: mala(t, sel("nothing"), 0.1))
    assert torch.equal(same.traces.get_choices()["x"], p.traces.get_choices()["x"])


def test_mala_beta_bernoulli_convergence(G):
    """test_mcmc.py:853-903: p ~ beta(2, 5), obs = True, step 0.05; posterior Beta(3, 5).  512 chains x 400 steps
    instead of one long chain, at a tolerance 5 times tighter than the reference's 0.1 on the mean."""
    from genjax_amd import beta, const, flip, gen, seed, sel
    from genjax_amd.inference import chain, mala

    @gen
    def beta_bernoulli():
        p = beta(2.0, 5.0) @ "p"
        return flip(p) @ "obs"

    tr, _ = seed(beta_bernoulli.generate)(G.key(3), {"obs": True})
    res = seed(chain(lambda t: mala(t, sel("p"), 0.05)))(G.key(4), tr, const(400), burn_in=const(150),
                                                         n_chains=const(512))
    p = res.traces.get_choices()["p"]
    mean, var, _, _ = beta_bernoulli_posterior(True, 2.0, 5.0)
    assert abs(float(p.mean()) - mean) < 0.02 and abs(float(p.var()) - var) < 0.006, (float(p.mean()), float(p.var()))
    assert float(res.acceptance_rate) > 0.2


def test_mala_multiple_parameters(G):
    """test_mcmc.py:1252-1312 shape: two selected sites move together; line through 3 points, Gaussian posterior
    with a closed form (the same target as the two-site hmc test below)."""
    from genjax_amd import const, gen, normal, seed, sel
    from genjax_amd.inference import chain, mala

    xs = torch.tensor([-1.0, 0.0, 1.0], device="cuda")
    ys = [0.2, 0.9, 2.1]

    @gen
    def line():
        a = normal(0.0, 1.0) @ "a"
        b = normal(0.0, 1.0) @ "b"
        for i in range(3):
            normal(a * xs[i] + b, 0.5) @ f"y{i}"

    tr, _ = seed(line.generate)(G.key(20), {f"y{i}": ys[i] for i in range(3)})
    res = seed(chain(lambda t: mala(t, sel("a") | sel("b"), 0.25)))(G.key(21), tr, const(300), burn_in=const(100),
                                                                  n_chains=const(512))
    a, b = res.traces.get_choices()["a"], res.traces.get_choices()["b"]
    # posterior precision: prior 1 + sum x^2 / 0.25 = 9 for a, 1 + 3 / 0.25 = 13 for b (x sums to 0: independent)
    ma, mb = (sum(x * y for x, y in zip([-1.0, 0.0, 1.0], ys)) / 0.25) / 9.0, (sum(ys) / 0.25) / 13.0
    assert abs(float(a.mean()) - ma) < 0.02 and abs(float(b.mean()) - mb) < 0.02, (float(a.mean()), float(b.mean()))
    assert abs(float(a.var()) - 1 / 9.0) < 0.01 and abs(float(b.var()) - 1 / 13.0) < 0.01
    assert float(res.acceptance_rate) > 0.5


def test_mala_refuses_fused_traces(G):
    from genjax_amd import sel
    from genjax_amd._native import GjxError
    from genjax_amd.families.base import ProbeTrace
    from genjax_amd.inference import mala

    with pytest.raises(GjxError, match="mala"):
        mala(ProbeTrace(None), sel("x"), 0.1)


def test_hmc_two_selected_sites_in_rejuvenate(G):
    """hmc over two leaves of a batched trace inside rejuvenate (the SMC use): line through 3 points, posterior
    of (a, b) is Gaussian with a closed form."""
    from genjax_amd import const, gen, normal, seed, sel
    from genjax_amd.inference import hmc, init, rejuvenate

    xs = torch.tensor([-1.0, 0.0, 1.0], device="cuda")
    ys = torch.tensor([0.2, 0.9, 2.1], device="cuda")

    @gen
    def point(x, a, b):
        return normal(a + b * x, 0.5) @ "obs"

    @gen
    def line(xs):
        a = normal(0.0, 1.0) @ "a"
        b = normal(0.0, 1.0) @ "b"
        return point.vmap(in_axes=(0, None, None))(xs, a, b) @ "ys"

    n = 20000
    p = seed(init)(G.key(7), line, (xs,), const(n), {"ys": {"obs": ys}})

    def moves(p):
        for _ in range(30):
            # trajectory length 0.44 = a quarter period of the posterior of a (sd 0.277): fast mixing
            p = rejuvenate(p, lambda t: hmc(t, sel("a") | sel("b"), 0.11, 4))
        return p

    p = seed(moves)(G.key(8), p)
    ch = p.traces.get_choices()
    X = np.stack([np.ones(3), xs.cpu().numpy()], axis=1)
    prec = np.eye(2) + X.T @ X / 0.25
    cov = np.linalg.inv(prec)
    m = cov @ (X.T @ ys.cpu().numpy() / 0.25)
    # the particles entered as prior draws with weights; 30 HMC moves leave the posterior invariant and move every
    # particle towards it: the unweighted cloud is at the posterior after mixing
    assert abs(float(ch["a"].mean()) - m[0]) < 0.03 and abs(float(ch["b"].mean()) - m[1]) < 0.03
    assert abs(float(ch["a"].var()) - cov[0, 0]) < 0.02 and abs(float(ch["b"].var()) - cov[1, 1]) < 0.02


# ------------------------------------------------------------------ chain() over fused traces
def test_chain_drives_fused_traces(G, oracle):
    """chain(kernel) on a fused-family trace (curvefit/core.py:164-210 drives its HMC through chain): the trace's
    particles are the chains, every step is one fused launch, and step i equals the kernel run alone on
    fold_in(scan key, i)."""
    from genjax_amd import const, seed, sel
    from genjax_amd.families.localization import Pose, create_basic_multi_room_world, localization_model
    from genjax_amd.inference import chain, init, mh

    ow = oracle.basic_world()
    obs_o, k = oracle.localization_observations(ow)
    world = create_basic_multi_room_world()
    n = 2048
    p = seed(init)(k, localization_model, (Pose(0.0, 0.0, 0.0), 0, world, const(8)), const(n),
                   {"measurements": {"distance": obs_o[0]}})
    s = sel("x") | sel("y") | sel("theta")
    kern = lambda t: mh(t, s)                       # noqa: E731
    res = seed(chain(kern))(G.key(9), p.traces, const(6), burn_in=const(2))
    x = res.traces.get_choices()["x"]
    assert x.shape == (n, 4) and res.n_chains.value == n and res.n_steps.value == 4
    assert 0.0 < float(res.acceptance_rate) < 1.0
    assert set(res.rhat) >= {"x", "y", "theta"}
    # step 0 of the scan = the kernel alone under fold_in(split(key)[1]-style scan key, 0)
    sub = G.backend.KeySeq(G.key(9)).next()
    first = seed(kern)(G.fold_in(sub, 0), p.traces)
    res_all = seed(chain(kern))(G.key(9), p.traces, const(1))
    assert torch.equal(res_all.traces.get_choices()["x"][:, 0], first.get_choices()["x"])
    with pytest.raises(G.GjxError):
        chain(kern)(p.traces, const(2), n_chains=const(3))


# ------------------------------------------------------------------ ParticleCollection arithmetic through the library
def test_particle_collection_weights_and_estimates(G):
    """diagnostic_weights = logw - lse (smc.py:259-261) and estimate() = sum softmax(logw) f (smc.py:206-245)
    against f64; both are library launches."""
    from genjax_amd import const, gen, normal, seed
    from genjax_amd.inference import init

    @gen
    def model():
        mu = normal(0.0, 1.0) @ "mu"
        return normal(mu, 0.5) @ "y"

    n = 300000
    p = seed(init)(G.key(0), model, (), const(n), {"y": 1.5})
    lw = p.log_weights.double()
    lse = torch.logsumexp(lw, 0)
    np.testing.assert_allclose(p.diagnostic_weights.cpu().numpy(), (lw - lse).float().cpu().numpy(), rtol=1e-5, atol=1e-5)
    w = torch.exp(lw - lse)
    mu = p.traces.get_choices()["mu"].double()
    est = p.estimate(lambda ch: ch["mu"])
    np.testing.assert_allclose(float(est), float((w * mu).sum()), rtol=1e-5)
    assert abs(float(est) - 1.2) < 0.01
    est2 = p.estimate(lambda ch: torch.stack([ch["mu"], ch["mu"] ** 2], dim=1))
    np.testing.assert_allclose(est2.cpu().numpy(), [float((w * mu).sum()), float((w * mu * mu).sum())], rtol=1e-5)
