"""Datasets of the curve-fitting case study (reference examples/curvefit/data.py:14-306).  Every draw is
`jax.random` on a key chain (`key, subkey = split(key)`), which the library's Threefry reproduces bit for bit:
the same seed gives the reference's dataset (up to the last-ulp differences of the normal transform documented
in DESIGN.md section 3.2)."""
from __future__ import annotations

import numpy as np
import torch

from .. import backend as B


def polyfn(x, a, b, c):
    """a + b x + c x^2 (data.py:14-27)."""
    return a + b * x + c * x**2


class _KeyChain:
    """key, subkey = jrand.split(key), repeated."""

    def __init__(self, key):
        self.key = np.asarray(key, np.uint32).reshape(2)

    def next(self):
        self.key, sub = B.split(self.key, 2)
        return sub

    def normal(self, shape=()):
        n = int(np.prod(shape)) if shape else 1
        out = B.sample_normal(self.next(), (n,), 0.0, 1.0)
        return out.reshape(shape)

    def uniform(self, shape=()):
        n = int(np.prod(shape)) if shape else 1
        return B.sample_uniform(self.next(), (n,), 0.0, 1.0).reshape(shape)


def _xs(lo, hi, n):
    return torch.linspace(lo, hi, n, dtype=torch.float32, device=B._dev())


def generate_test_dataset(key=None, n_points=10, seed=42):
    """xs on [0, 1] and ys simulated from npoint_curve under the key (data.py:30-89)."""
    from ..core import seed as seed_
    from .curvefit import npoint_curve_generic

    key = B.key(seed) if key is None else key
    xs = _xs(0, 1, n_points)
    trace = seed_(npoint_curve_generic.simulate)(key, xs)
    _, (_, ys) = trace.get_retval()
    cur = trace.get_choices()["curve"]
    a, b, c = (float(cur[k]) for k in "abc")
    return {"xs": xs, "ys": ys, "true_params": {"a": a, "b": b, "c": c, "noise_std": 0.05},
            "clean_ys": polyfn(xs, a, b, c), "trace": trace}


def generate_easy_inference_dataset(seed=42, n_points=5, noise_std=0.05, param_scale=0.3, key=None):
    """Few points and coefficients near the prior mean (data.py:92-154)."""
    chain = _KeyChain(B.key(seed) if key is None else key)
    xs = _xs(0, 1, n_points)
    a, b, c = (param_scale * float(chain.normal()) for _ in range(3))
    clean = polyfn(xs, a, b, c)
    ys = clean + noise_std * chain.normal((n_points,))
    return {"xs": xs, "ys": ys, "true_params": {"a": a, "b": b, "c": c, "noise_std": noise_std}, "clean_ys": clean,
            "n_points": n_points}


def generate_fixed_dataset(n_points=10, x_min=0.0, x_max=1.0, true_a=-0.211, true_b=-0.395, true_c=0.673,
                           noise_std=0.05, seed=42):
    """Fixed coefficients, seeded noise (data.py:157-206)."""
    chain = _KeyChain(B.key(seed))
    xs = _xs(x_min, x_max, n_points)
    clean = polyfn(xs, true_a, true_b, true_c)
    ys = clean + noise_std * chain.normal((n_points,))
    return {"xs": xs, "ys": ys, "true_params": {"a": true_a, "b": true_b, "c": true_c}, "noise_std": noise_std,
            "clean_ys": clean}


def generate_dataset_with_outliers(key=None, n_points=20, outlier_fraction=0.15, outlier_scale=3.0, noise_std=0.05,
                                   seed=42, true_a=None, true_b=None, true_c=None):
    """Inlier noise noise_std, outlier noise noise_std * outlier_scale on a uniform-thresholded subset
    (data.py:209-292)."""
    chain = _KeyChain(B.key(seed) if key is None else key)
    xs = _xs(0, 1, n_points)
    coeffs = [float(chain.normal()) * 0.5 if v is None else v for v in (true_a, true_b, true_c)]
    a, b, c = coeffs
    clean = polyfn(xs, a, b, c)
    is_outlier = chain.uniform((n_points,)) < outlier_fraction
    inlier_noise = noise_std * chain.normal((n_points,))
    outlier_noise = noise_std * outlier_scale * chain.normal((n_points,))
    ys = clean + torch.where(is_outlier, outlier_noise, inlier_noise)
    return {"xs": xs, "ys": ys,
            "true_params": {"a": float(a), "b": float(b), "c": float(c), "noise_std": noise_std,
                            "outlier_fraction": outlier_fraction, "outlier_scale": outlier_scale},
            "clean_ys": clean, "is_outlier": is_outlier, "n_outliers": int(is_outlier.sum())}


def get_reference_dataset(seed=42, n_points=20):
    """The dataset every figure of the case study uses (data.py:295-306)."""
    return generate_fixed_dataset(n_points=n_points, x_min=0.0, x_max=1.0, true_a=-0.211, true_b=-0.395, true_c=0.673,
                                  noise_std=0.05, seed=seed)


def print_dataset_summary(data_dict, name="Dataset"):
    """data.py:309-333."""
    print(f"\n{name} Summary:")
    print(f"  Number of points: {len(data_dict['xs'])}")
    print(f"  X range: [{float(data_dict['xs'].min()):.3f}, {float(data_dict['xs'].max()):.3f}]")
    print(f"  Y range: [{float(data_dict['ys'].min()):.3f}, {float(data_dict['ys'].max()):.3f}]")
    tp = data_dict["true_params"]
    print(f"  True parameters: a={tp['a']:.3f}, b={tp['b']:.3f}, c={tp['c']:.3f}")
    if "n_outliers" in data_dict:
        print(f"  Outliers: {data_dict['n_outliers']}")
