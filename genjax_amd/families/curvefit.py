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
        from ..core import batch_size

        return self.fused_init(args, batch_size() or 1, x)


npoint_curve = NPointCurve()


def infer_latents(xs, ys, n_samples: Const):
    """Importance sampling of the curve parameters (core.py:114-136)."""
    from ..inference import init

    result = init(npoint_curve, (xs,), n_samples, {"ys": {"obs": ys}})
    return result.traces, result.log_weights


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
