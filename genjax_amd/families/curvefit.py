"""Curve-fitting case study (reference examples/curvefit/core.py:57-136).

`npoint_curve` is the fused family behind `init(npoint_curve, (xs,), n, {"ys": {"obs": ys}})`
(one kernel: three Threefry normals + the 8-point Gaussian likelihood, 16 B written per particle).
`npoint_curve_generic` is the same model written with `@gen`, running on the generic column path;
the two are cross-checked in the GPU tests.

The outlier-mixture models (core.py:468-552) are written with `@gen` for the generic column path
(`point_with_outliers`, `npoint_curve_with_outliers[_beta]`, `Cond` = both branches + select,
core.py:2638-2785), and the per-chain enumerative-Gibbs + HMC kernel of core.py:934-1014 /
figs.py:907-950 is one fused launch (`run_gibbs_hmc`, gjx_outlier_gibbs_hmc)."""
from __future__ import annotations

import numpy as np
import torch

from .. import backend as B
from ..core import Cond, Const, gen
from ..distributions import normal
from .._native import GjxError
from .base import FusedFamily, FusedTrace


def polyfn(x, coeffs):
    """Degree-2 polynomial a + b x + c x^2 (core.py:72-75); coeffs = (a, b, c) columns."""
    a, b, c = coeffs
    return a + b * x + c * x**2


class Lambda:
    """Curve closure (core.py:44-53)."""

    def __init__(self, f, dynamic_vals):
        self.f, self.dynamic_vals = f, dynamic_vals

    def __call__(self, x):
        return self.f(x, self.dynamic_vals)

    def unsqueeze_params(self):
        return Lambda(self.f, tuple(v.unsqueeze(-1) if isinstance(v, torch.Tensor) and v.dim() >= 1 else v
                                    for v in self.dynamic_vals))


@gen
def point(x, curve):
    y_det = curve(x)
    return normal(y_det, 0.05) @ "obs"


@gen
def polynomial():
    a = normal(0.0, 1.0) @ "a"
    b = normal(0.0, 1.0) @ "b"
    c = normal(0.0, 1.0) @ "c"
    return Lambda(polyfn, (a, b, c))


@gen
def npoint_curve_generic(xs):
    curve = polynomial() @ "curve"
    ys = point.vmap(in_axes=(0, None))(xs, curve) @ "ys"
    return curve, (xs, ys)


@gen
def point_with_noise(x, curve, noise_std):
    """Point model with a configurable noise level (core.py:64-69)."""
    y_det = curve(x)
    return normal(y_det, noise_std) @ "obs"


@gen
def onepoint_curve(x):
    curve = polynomial() @ "curve"
    y = point(x, curve) @ "y"
    return curve, (x, y)


@gen
def npoint_curve_easy(xs, noise_std=Const(0.2)):
    """npoint_curve with more observation noise, for easier inference (core.py:103-111)."""
    curve = polynomial() @ "curve"
    sd = noise_std.value if isinstance(noise_std, Const) else noise_std
    ys = point_with_noise.vmap(in_axes=(0, None, None))(xs, curve, sd) @ "ys"
    return curve, (xs, ys)


class CurveTrace(FusedTrace):
    def __init__(self, family, a, b, c, xs, ys, sigma):
        self.family, self.a, self.b, self.c = family, a, b, c
        self.xs, self.ys, self.sigma = xs, ys, sigma

    @property
    def x(self):
        return self.a

    def columns(self):
        return {"a": self.a, "b": self.b, "c": self.c}

    def gather(self, ancestors, idx_base=0):
        a, b, c = B.gather_columns(ancestors, [self.a, self.b, self.c], idx_base=idx_base)
        return CurveTrace(self.family, a, b, c, self.xs, self.ys, self.sigma)

    def get_choices(self):
        ys = torch.as_tensor(self.ys, device=self.a.device).expand(self.a.numel(), -1)
        return {"curve": {"a": self.a, "b": self.b, "c": self.c}, "ys": {"obs": ys}}

    def get_retval(self):
        return Lambda(polyfn, (self.a, self.b, self.c)), (self.xs, self.ys)

    def get_args(self):
        return ((self.xs,), {})


class NPointCurve(FusedFamily):
    """npoint_curve(xs) (core.py:95-100) with observation noise 0.05 (core.py:60)."""

    name = "npoint_curve"
    sigma = 0.05

    def fused_init(self, args, n, constraints, offset=0):
        from ..core import _next_key

        (xs,) = args
        try:
            ys = constraints["ys"]["obs"]
        except (KeyError, TypeError):
            raise GjxError('npoint_curve: constraints must be {"ys": {"obs": ys}}')
        xs = xs.detach().cpu().numpy() if isinstance(xs, torch.Tensor) else np.asarray(xs)
        ys = ys.detach().cpu().numpy() if isinstance(ys, torch.Tensor) else np.asarray(ys)
        keys = [_next_key() for _ in range(3)]            # curve.a, curve.b, curve.c (core.py:82-84)
        a, b, c, logw = B.poly_is(keys, n, xs, ys, self.sigma, offset=offset)
        return CurveTrace(self, a, b, c, xs.astype(np.float32), ys.astype(np.float32), self.sigma), logw

    def generate(self, x, *args, **kwargs):
        """Fused when only the observations are constrained; the @gen twin when a coefficient is (the generate
        call behind a proposal, smc.py:332-350)."""
        from ..core import batch_size

        if set(x or {}) != {"ys"}:
            return self.generic.generate(x, *args, **kwargs)
        return self.fused_init(args, batch_size() or 1, x)


npoint_curve = NPointCurve()
npoint_curve.generic = npoint_curve_generic


def infer_latents(xs, ys, n_samples: Const):
    """Importance sampling of the curve parameters (core.py:114-136)."""
    from ..inference import init

    result = init(npoint_curve, (xs,), n_samples, {"ys": {"obs": ys}})
    return result.traces, result.log_weights


def infer_latents_easy(xs, ys, n_samples: Const, noise_std=0.2):
    """Importance sampling under npoint_curve_easy (core.py:139-161)."""
    from ..inference import init

    result = init(npoint_curve_easy, (xs, noise_std), n_samples, {"ys": {"obs": ys}})
    return result.traces, result.log_weights


def _chain_diagnostics(result, full=False):
    d = {"acceptance_rate": result.acceptance_rate, "n_samples": result.n_steps.value,
         "n_chains": result.n_chains.value}
    if full:
        d.update(rhat=result.rhat, ess_bulk=result.ess_bulk, ess_tail=result.ess_tail)
    return d


def hmc_infer_latents(xs, ys, n_samples: Const, n_warmup: Const = Const(500), step_size=0.05,
                      n_steps: Const = Const(10), n_chains: Const = Const(1)):
    """HMC over the curve coefficients (core.py:164-210): generate an initial trace under the observations,
    then chain(hmc(sel("curve"))) with n_warmup burn-in steps.  -> (traces, diagnostics).  The model is the
    @gen npoint_curve (generic column path: log-density and gradient kernels under autograd)."""
    from ..core import sel
    from ..inference import chain, hmc

    initial_trace, _ = npoint_curve_generic.generate({"ys": {"obs": ys}}, xs)
    L = n_steps.value if isinstance(n_steps, Const) else int(n_steps)
    run = chain(lambda trace: hmc(trace, sel("curve"), step_size=step_size, n_steps=L))
    result = run(initial_trace, Const(n_samples.value + n_warmup.value), burn_in=n_warmup, n_chains=n_chains)
    return result.traces, _chain_diagnostics(result)


def hmc_infer_latents_vectorized(xs, ys, n_samples: Const, n_warmup: Const = Const(500), step_size=0.05,
                                 n_steps: Const = Const(10), n_chains: Const = Const(4)):
    """n_chains chains at once (core.py:213-254); leaves are (n_chains, n_samples).  The reference vmaps the
    single-chain program over split(key(0), n_chains); here the chains are the particle axis of one program
    (chain(..., n_chains=)), which draws each step's column from that step's key."""
    return hmc_infer_latents(xs, ys, n_samples, n_warmup, step_size, n_steps, n_chains=n_chains)


def get_points_for_inference(n_points=10):
    """xs on a grid in [0, 1] and ys simulated from the model (core.py:257-264)."""
    xs = torch.linspace(0, 1, n_points, device=B._dev())
    _, (_, ys) = npoint_curve_generic.simulate(xs).get_retval()
    return xs, ys


def log_marginal_likelihood(log_weights):
    """logsumexp(logw) - log n (core.py:267-269)."""
    return B.logsumexp_ess(log_weights)[1] - float(np.log(log_weights.numel()))


def effective_sample_size(log_weights):
    """1 / sum softmax(logw)^2 (core.py:272-276)."""
    return B.logsumexp_ess(log_weights)[2]


# --------------------------------------------------------------------------- #
# outlier-mixture models (core.py:468-552)
# --------------------------------------------------------------------------- #
def _import_outlier_dists():
    from ..distributions import beta, flip, uniform

    return beta, flip, uniform


@gen
def inlier_branch(mean):
    return normal(mean, 0.1) @ "obs"


@gen
def outlier_branch(mean):
    _, _, uniform = _import_outlier_dists()
    del mean
    return uniform(-2.0, 2.0) @ "obs"


@gen
def point_with_outliers(x, curve, outlier_rate=0.1):
    _, flip, _ = _import_outlier_dists()
    y_det = curve(x)
    is_outlier = flip(outlier_rate) @ "is_outlier"
    return Cond(outlier_branch, inlier_branch)(is_outlier, y_det) @ "y"


@gen
def npoint_curve_with_outliers(xs, outlier_rate=0.1):
    curve = polynomial() @ "curve"
    ys = point_with_outliers.vmap(in_axes=(0, None, None))(xs, curve, outlier_rate) @ "ys"
    return curve, (xs, ys)


@gen
def npoint_curve_with_outliers_beta(xs, alpha=1.0, beta_param=10.0):
    beta, _, _ = _import_outlier_dists()
    outlier_rate = beta(alpha, beta_param) @ "outlier_rate"
    curve = polynomial() @ "curve"
    ys = point_with_outliers.vmap(in_axes=(0, None, None))(xs, curve, outlier_rate) @ "ys"
    return curve, (xs, ys)


def infer_latents_with_outliers(xs, ys, n_samples: Const, outlier_rate=0.1):
    """Importance sampling of curve and outlier indicators (core.py:555-583)."""
    from ..inference import init

    result = init(npoint_curve_with_outliers, (xs, outlier_rate), n_samples, {"ys": {"y": {"obs": ys}}})
    return result.traces, result.log_weights


def hmc_infer_latents_with_outliers(xs, ys, n_samples: Const, n_warmup: Const = Const(500), step_size=0.05,
                                    n_steps: Const = Const(10), outlier_rate=0.1, n_chains: Const = Const(1)):
    """HMC over the curve coefficients with the outlier indicators held at their initial draw (core.py:586-628)."""
    from ..core import sel
    from ..inference import chain, hmc

    initial_trace, _ = npoint_curve_with_outliers.generate({"ys": {"y": {"obs": ys}}}, xs, outlier_rate)
    L = n_steps.value if isinstance(n_steps, Const) else int(n_steps)
    run = chain(lambda trace: hmc(trace, sel("curve"), step_size=step_size, n_steps=L))
    result = run(initial_trace, Const(n_samples.value + n_warmup.value), burn_in=n_warmup, n_chains=n_chains)
    return result.traces, _chain_diagnostics(result)


def enumerative_gibbs_outliers(trace, xs, ys, outlier_rate=0.1):
    """Exact Gibbs update of every outlier indicator (core.py:934-985): per point the two joint log-densities
    log p(is_outlier = b, y | curve, rate), one categorical draw over them, then one update of the trace.
    The rate is the trace's own "outlier_rate" site when the model has one (the beta model)."""
    from ..core import as_col, batch_size
    from ..distributions import categorical

    ch = trace.get_choices()
    cp = ch["curve"]
    rate = ch["outlier_rate"] if "outlier_rate" in ch else as_col(outlier_rate)
    n = batch_size()
    xs_, ys_ = as_col(xs), as_col(ys)
    lift = (lambda v: v.unsqueeze(-1) if isinstance(v, torch.Tensor) and n is not None and v.dim() >= 1 else v)
    curve = Lambda(polyfn, tuple(lift(cp[k]) for k in ("a", "b", "c")))
    rate_ = lift(as_col(rate))
    zeros, ones = torch.zeros_like(xs_), torch.ones_like(xs_)
    lp = []
    for flag in (zeros, ones):                                    # both branches for every point at once
        # point_with_outliers.assess over the points axis: flip(rate) at the flag + the branch's density of y
        y_det = curve(xs_)
        _, flip, uniform = _import_outlier_dists()
        lp_flag, _ = flip.assess(flag, rate_)
        lp_obs, _ = (uniform.assess(ys_, -2.0, 2.0) if flag is ones else normal.assess(ys_, y_det, 0.1))
        lp.append(lp_flag + lp_obs)
    logits = torch.stack(lp, dim=-1)                              # (..., n_points, 2)
    flat = categorical.sample(logits.reshape(-1, 2))
    new_outliers = flat.reshape(logits.shape[:-1]).to(torch.float32)
    gen_fn, args = trace.get_gen_fn(), trace.get_args()
    new_trace, _, _ = gen_fn.update(trace, {"ys": {"is_outlier": new_outliers}}, *args[0], **args[1])
    return new_trace


def mixed_gibbs_hmc_kernel(xs, ys, hmc_step_size=0.01, hmc_n_steps=10, outlier_rate=0.1):
    """kernel(trace): enumerative Gibbs over the indicators, then HMC over (a, b, c, outlier_rate)
    (core.py:988-1014).  The fused form of the same kernel for many chains is run_gibbs_hmc below."""
    from ..core import sel
    from ..inference import hmc

    L = hmc_n_steps.value if isinstance(hmc_n_steps, Const) else int(hmc_n_steps)

    def kernel(trace):
        trace = enumerative_gibbs_outliers(trace, xs, ys, outlier_rate)
        selection = sel(("curve", "a")) | sel(("curve", "b")) | sel(("curve", "c")) | sel("outlier_rate")
        return hmc(trace, selection, step_size=hmc_step_size, n_steps=L)

    return kernel


def mixed_infer_latents_with_outliers_beta(xs, ys, n_samples: Const, n_warmup: Const = Const(500),
                                           hmc_step_size=0.01, hmc_n_steps: Const = Const(10), alpha=1.0,
                                           beta_param=10.0, n_chains: Const = Const(1)):
    """Mixed MCMC on the beta-rate outlier model (core.py:631-691): Gibbs for the indicators, HMC for the
    continuous sites; -> (traces, diagnostics with rhat / ess when several chains run)."""
    from ..inference import chain

    initial_trace, _ = npoint_curve_with_outliers_beta.generate({"ys": {"y": {"obs": ys}}}, xs, alpha, beta_param)
    run = chain(mixed_gibbs_hmc_kernel(xs, ys, hmc_step_size=hmc_step_size, hmc_n_steps=hmc_n_steps))
    result = run(initial_trace, Const(n_samples.value + n_warmup.value), burn_in=n_warmup, n_chains=n_chains)
    return result.traces, _chain_diagnostics(result, full=True)


class GibbsHmcChains:
    """Result of `run_gibbs_hmc`: per-chain final state on the device."""

    def __init__(self, model, state, n_points, accepts, n_moves):
        self.model, self.state, self.n_points = model, state, n_points
        self.accept_rate = float(accepts) / max(1, n_moves)

    @property
    def is_outlier(self) -> torch.Tensor:
        """(n_chains, n_points) bool: choices["ys"]["is_outlier"]."""
        bits = torch.arange(self.n_points, device=self.state["outliers"].device, dtype=torch.int64)
        return ((self.state["outliers"][:, None] >> bits[None, :]) & 1).bool()

    def get_choices(self):
        s = self.state
        return {"outlier_rate": s["rate"], "curve": {"a": s["a"], "b": s["b"], "c": s["c"]},
                "ys": {"is_outlier": self.is_outlier}}

    def outlier_probabilities(self) -> torch.Tensor:
        """Posterior P(is_outlier_j) estimated over chains (figs.py:944)."""
        return self.is_outlier.float().mean(dim=0)


def run_gibbs_hmc(xs, ys, key, n_chains: int, n_iterations: int, hmc_step_size: float = 0.01,
                  hmc_n_steps: int = 20, alpha: float = 1.0, beta_param: float = 3.0, state=None,
                  offset: int = 0) -> GibbsHmcChains:
    """The reference's `run_gibbs_hmc` (figs.py:907-950): `n_chains` independent chains with keys
    `split(key, n_chains)`, each initialised by generate() of npoint_curve_with_outliers_beta
    (alpha, beta_param) under its own key and advanced by `n_iterations` applications of
    mixed_gibbs_hmc_kernel (core.py:988-1014) -- all iterations in ONE fused launch.
    Pass `state` (a previous result's .state) to continue chains."""
    model = B.make_outlier_model(xs, ys, alpha, beta_param)
    if state is None:
        state = B.outlier_init(model, key, n_chains, offset=offset)
    acc = torch.zeros(1, dtype=torch.int64, device=state["a"].device)
    B.outlier_gibbs_hmc(model, state, n_iterations, hmc_step_size, hmc_n_steps, accept_count=acc)
    return GibbsHmcChains(model, state, len(np.asarray(xs).reshape(-1)), int(acc.item()),
                          n_chains * max(1, n_iterations))


# the seeded entry points of the case study (core.py:694-701)
def _seeded(f):
    from ..core import seed

    return seed(f)


infer_latents_seeded = _seeded(infer_latents)
hmc_infer_latents_seeded = _seeded(hmc_infer_latents)
hmc_infer_latents_vectorized_seeded = _seeded(hmc_infer_latents_vectorized)
infer_latents_with_outliers_seeded = _seeded(infer_latents_with_outliers)
hmc_infer_latents_with_outliers_seeded = _seeded(hmc_infer_latents_with_outliers)
mixed_infer_latents_with_outliers_beta_seeded = _seeded(mixed_infer_latents_with_outliers_beta)
