"""Curve-fitting case study (reference examples/curvefit/core.py:57-136).

`npoint_curve` is the fused family behind `init(npoint_curve, (xs,), n, {"ys": {"obs": ys}})`
(one kernel: three Threefry normals + the 8-point Gaussian likelihood, 16 B written per particle).
`npoint_curve_generic` is the same model written with `@gen`, running on the generic column path;
the two are cross-checked in the GPU tests."""
from __future__ import annotations

import numpy as np
import torch

from .. import backend as B
from ..core import Const, gen
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
