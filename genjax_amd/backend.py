"""Thin host layer over the C ABI: torch-ROCm tensors are used ONLY as HBM buffers
(`tensor.data_ptr()`), streams come from torch's current stream, all arithmetic
happens in libgjx_hip.so.  No CPU fallback exists: without a GPU or without the
library every call raises.
"""
from __future__ import annotations

import collections
import ctypes as C
import math
from typing import Sequence

import numpy as np
import torch

from . import _native as N
from ._native import GjxError, LgssmParams, LocWorld, OutlierModel

# --------------------------------------------------------------------------- #
# host-side key arithmetic (jax.random.key / split / fold_in)
# --------------------------------------------------------------------------- #
_U32x2 = C.c_uint32 * 2


def _ckey(k):
    k = np.asarray(k, dtype=np.uint32).reshape(2)
    return _U32x2(int(k[0]), int(k[1]))


def key(seed: int) -> np.ndarray:
    """jax.random.key(seed) as two uint32 words (hi, lo)."""
    out = _U32x2()
    N.lib().gjx_key(int(seed), out)
    return np.array([out[0], out[1]], dtype=np.uint32)


def split(k, n: int = 2) -> np.ndarray:
    """jax.random.split(k, n) -> (n, 2) uint32."""
    out = (C.c_uint32 * (2 * n))()
    N.lib().gjx_split(_ckey(k), n, out)
    return np.frombuffer(out, dtype=np.uint32).reshape(n, 2).copy()


def fold_in(k, data: int) -> np.ndarray:
    out = _U32x2()
    N.lib().gjx_fold_in(_ckey(k), int(data), out)
    return np.array([out[0], out[1]], dtype=np.uint32)


class KeySeq:
    """State of the reference's Seed interpreter (pjax.py:1316,1337): every sampling site,
    cond and scan takes `self.key, sub = split(self.key)`."""

    def __init__(self, k):
        self.key = np.asarray(k, dtype=np.uint32).reshape(2).copy()

    def next(self) -> np.ndarray:
        ks = split(self.key, 2)
        self.key = ks[0]
        return ks[1]


# --------------------------------------------------------------------------- #
# device plumbing
# --------------------------------------------------------------------------- #
def _require_gpu():
    if not torch.cuda.is_available():
        raise GjxError("genjax_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                       "there is no CPU fallback")


def _ptr(t: torch.Tensor | None):
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise GjxError("expected a device tensor")
    if not t.is_contiguous():
        raise GjxError("expected a contiguous tensor")
    return C.c_void_p(t.data_ptr())


def _rows(t: torch.Tensor | None):
    """(pointer, row stride) of a 2-D float tensor whose rows are contiguous (rows may be spaced)."""
    if t is None:
        return C.c_void_p(0), 0
    if not t.is_cuda or t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.float32:
        raise GjxError("expected a 2-D float32 device tensor with contiguous rows")
    return C.c_void_p(t.data_ptr()), int(t.stride(0))


def _raw_stream() -> int:
    """The current device's current stream as the raw hipStream_t (torch.cuda.current_stream() builds a Stream
    object and resolves the device through several Python layers: ~9 us, eight times per SMC stage)."""
    return _RAW_STREAM(torch.cuda.current_device())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (
    lambda index: torch.cuda.current_stream(index).cuda_stream)


def _stream():
    return C.c_void_p(_raw_stream())


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise GjxError(f"expected float32, got {t.dtype}")
    return t


class Context:
    """A gjx_ctx (reduction / scan workspace) per (device, stream): launches that share a workspace must be
    ordered by ONE stream, so two SMC instances on two streams of a device get two contexts.  A context
    replaced by a larger one is destroyed when the last buffer that references it is dropped.  The cache holds the
    MAX_CACHED most recently used contexts: a program that keeps making new streams drops the oldest workspace
    instead of pinning one per stream for the life of the process (a dropped context whose stream comes back is
    simply made again; destroying one synchronises, so nothing in flight loses its workspace)."""

    MAX_CACHED = 8
    _cache: "collections.OrderedDict" = collections.OrderedDict()

    def __init__(self, device: int, n_max: int):
        _require_gpu()
        self.device, self.n_max = device, int(n_max)
        self.handle = None
        h = C.c_void_p()
        N.check(N.lib().gjx_ctx_create(device, self.n_max, C.byref(h)), "gjx_ctx_create")
        self.handle = h

    def __del__(self):
        try:
            if self.handle is not None:
                N.lib().gjx_ctx_destroy(self.handle)      # hipFree synchronises: no launch still uses the workspace
                self.handle = None
        except Exception:                                 # interpreter shutdown: the library may be gone already
            pass

    @classmethod
    def get(cls, device: int, n: int) -> "Context":
        if device != torch.cuda.current_device():
            raise GjxError(f"tensors live on cuda:{device} but the current device is cuda:{torch.cuda.current_device()}: "
                           "launches go to the current device's stream (use torch.cuda.set_device / torch.cuda.device)")
        key_ = (device, _raw_stream())
        c = cls._cache.get(key_)
        if c is None or c.n_max < n:
            c = cls(device, max(int(n), 1 << 16))
            cls._cache[key_] = c
            while len(cls._cache) > cls.MAX_CACHED:
                cls._cache.popitem(last=False)
        cls._cache.move_to_end(key_)
        return c


def _ctx_for(t: torch.Tensor) -> Context:
    return Context.get(t.device.index or 0, t.numel())


def _dev(device=None) -> torch.device:
    _require_gpu()
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device(device)


# --------------------------------------------------------------------------- #
# draws
# --------------------------------------------------------------------------- #
def random_bits(k, n: int, offset: int = 0, device=None) -> torch.Tensor:
    """jax.random.bits(k, (n,), uint32) returned as int32 storage (same bit pattern)."""
    out = torch.empty(n, dtype=torch.int32, device=_dev(device))
    N.check(N.lib().gjx_random_bits_u32(_ckey(k), offset, n, _ptr(out), _stream()), "gjx_random_bits_u32")
    return out


def random_uniform(k, n: int, minval: float = 0.0, maxval: float = 1.0, offset: int = 0,
                   device=None) -> torch.Tensor:
    out = torch.empty(n, dtype=torch.float32, device=_dev(device))
    N.check(N.lib().gjx_random_uniform_f32(_ckey(k), offset, n, minval, maxval, _ptr(out), _stream()),
            "gjx_random_uniform_f32")
    return out


def random_normal(k, n: int, loc: float = 0.0, scale: float = 1.0, offset: int = 0,
                  device=None) -> torch.Tensor:
    out = torch.empty(n, dtype=torch.float32, device=_dev(device))
    N.check(N.lib().gjx_random_normal_f32(_ckey(k), offset, n, loc, scale, _ptr(out), _stream()),
            "gjx_random_normal_f32")
    return out


def _col_param(p, shape, dev):
    """Parameter broadcast against `shape`: (tensor, stride) with stride 0 for scalars."""
    n = int(np.prod(shape)) if len(shape) else 1
    if isinstance(p, torch.Tensor) and p.numel() > 1:
        return _f32(p).expand(shape).contiguous().view(-1), 1
    if isinstance(p, torch.Tensor):
        return _f32(p).reshape(1).contiguous(), (1 if n == 1 else 0)
    return torch.full((1,), float(p), dtype=torch.float32, device=dev), (1 if n == 1 else 0)


def sample_normal(k, shape, loc, scale, offset: int = 0, device=None) -> torch.Tensor:
    dev = _dev(device)
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    if out.numel() == 0:
        return out
    tl, sl = _col_param(loc, shape, dev)
    ts, ss = _col_param(scale, shape, dev)
    N.check(N.lib().gjx_random_normal_cols_f32(_ckey(k), offset, out.numel(), _ptr(tl), sl, _ptr(ts), ss,
                                               _ptr(out), _stream()), "gjx_random_normal_cols_f32")
    return out


def sample_uniform(k, shape, lo, hi, offset: int = 0, device=None) -> torch.Tensor:
    dev = _dev(device)
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    if out.numel() == 0:
        return out
    tl, sl = _col_param(lo, shape, dev)
    th, sh = _col_param(hi, shape, dev)
    N.check(N.lib().gjx_random_uniform_cols_f32(_ckey(k), offset, out.numel(), _ptr(tl), sl, _ptr(th), sh,
                                                _ptr(out), _stream()), "gjx_random_uniform_cols_f32")
    return out


def sample_bernoulli(k, shape, p, offset: int = 0, device=None) -> torch.Tensor:
    dev = _dev(device)
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    if out.numel() == 0:
        return out
    tp, sp = _col_param(p, shape, dev)
    N.check(N.lib().gjx_random_bernoulli_f32(_ckey(k), offset, out.numel(), _ptr(tp), sp, _ptr(out), _stream()),
            "gjx_random_bernoulli_f32")
    return out


def sample_beta(k, shape, a, b, offset: int = 0, device=None) -> torch.Tensor:
    dev = _dev(device)
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    if out.numel() == 0:
        return out
    ta, sa = _col_param(a, shape, dev)
    tb, sb = _col_param(b, shape, dev)
    N.check(N.lib().gjx_random_beta_f32(_ckey(k), offset, out.numel(), _ptr(ta), sa, _ptr(tb), sb, _ptr(out), _stream()),
            "gjx_random_beta_f32")
    return out


def sample_bernoulli_logits(k, shape, logits, offset: int = 0, device=None) -> torch.Tensor:
    dev = _dev(device)
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    if out.numel() == 0:
        return out
    tl, sl = _col_param(logits, shape, dev)
    N.check(N.lib().gjx_random_bernoulli_logits_f32(_ckey(k), offset, out.numel(), _ptr(tl), sl, _ptr(out), _stream()),
            "gjx_random_bernoulli_logits_f32")
    return out


def sample_categorical(k, logits: torch.Tensor, n: int | None = None, offset: int = 0) -> torch.Tensor:
    """logits (n, k): one draw per row; logits (k,): n draws from the shared row."""
    logits = _f32(logits).contiguous()
    kk = logits.shape[-1]
    if logits.dim() == 2:
        n, stride = logits.shape[0], 1
    else:
        n, stride = int(n or 1), 0
    out = torch.empty(n, dtype=torch.int32, device=logits.device)
    N.check(N.lib().gjx_random_categorical_i32(_ckey(k), offset, n, _ptr(logits), stride, kk, _ptr(out),
                                               _stream()), "gjx_random_categorical_i32")
    return out


def sample_mvnormal(k, n: int, mu: torch.Tensor, cov: torch.Tensor, offset: int = 0) -> torch.Tensor:
    mu = _f32(mu).contiguous()
    cov = _f32(cov).contiguous()
    d = mu.shape[-1]
    out = torch.empty((n, d), dtype=torch.float32, device=mu.device)
    N.check(N.lib().gjx_random_mvnormal_f32(_ckey(k), offset, n, d, _ptr(mu), 1 if mu.dim() == 2 else 0,
                                            _ptr(cov), 1 if cov.dim() == 3 else 0, _ptr(out), _stream()),
            "gjx_random_mvnormal_f32")
    return out


# --------------------------------------------------------------------------- #
# log densities
# --------------------------------------------------------------------------- #
def _param(p, like: torch.Tensor):
    """A distribution parameter as (tensor, stride): scalars broadcast with stride 0."""
    return _col_param(p, tuple(like.shape), like.device)


def _bshape(x, *params):
    shapes = [tuple(x.shape)] + [tuple(p.shape) for p in params if isinstance(p, torch.Tensor)]
    return tuple(torch.broadcast_shapes(*shapes))


def _elementwise2(fn_name, x, a, b):
    x = _f32(x)
    shp = _bshape(x, a, b)
    x = x.expand(shp).contiguous()
    out = torch.empty_like(x)
    if x.numel() == 0:
        return out
    ta, sa = _param(a, x)
    tb, sb = _param(b, x)
    fn = getattr(N.lib(), fn_name)
    N.check(fn(_ptr(x), _ptr(ta), sa, _ptr(tb), sb, x.numel(), _ptr(out), _stream()), fn_name)
    return out


def normal_logpdf(x, mu, sigma):
    return _elementwise2("gjx_normal_logpdf_f32", x, mu, sigma)


def uniform_logpdf(x, lo, hi):
    return _elementwise2("gjx_uniform_logpdf_f32", x, lo, hi)


def beta_logpdf(x, a, b):
    return _elementwise2("gjx_beta_logpdf_f32", x, a, b)


def bernoulli_logpdf(x, p):
    x = _f32(x)
    x = x.expand(_bshape(x, p)).contiguous()
    out = torch.empty_like(x)
    if x.numel() == 0:
        return out
    tp, sp = _param(p, x)
    N.check(N.lib().gjx_bernoulli_logpdf_f32(_ptr(x), _ptr(tp), sp, x.numel(), _ptr(out), _stream()),
            "gjx_bernoulli_logpdf_f32")
    return out


def bernoulli_logits_logpdf(x, logits):
    x = _f32(x)
    x = x.expand(_bshape(x, logits)).contiguous()
    out = torch.empty_like(x)
    if x.numel() == 0:
        return out
    tl, sl = _param(logits, x)
    N.check(N.lib().gjx_bernoulli_logits_logpdf_f32(_ptr(x), _ptr(tl), sl, x.numel(), _ptr(out), _stream()),
            "gjx_bernoulli_logits_logpdf_f32")
    return out


LOGPDF_KINDS = {"normal": 0, "uniform": 1, "beta": 2, "bernoulli": 3}


def logpdf_grad(kind: str, x, p1, p2, upstream, want=(True, True, True)):
    """upstream x d logpdf / d(x, p1, p2) over the broadcast shape of the inputs (entries not wanted: None)."""
    x = _f32(x)
    shp = _bshape(x, *(p for p in (p1, p2, upstream) if isinstance(p, torch.Tensor)))
    x = x.expand(shp).contiguous()
    t1, s1 = _param(p1, x)
    t2, s2 = _param(p2, x) if p2 is not None else (None, 0)
    tu, su = _param(upstream, x)
    outs = [torch.empty_like(x) if w else None for w in want]
    if x.numel() > 0:
        N.check(N.lib().gjx_logpdf_grad_f32(LOGPDF_KINDS[kind], _ptr(x), _ptr(t1), s1, _ptr(t2), s2, _ptr(tu), su,
                                            x.numel(), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _stream()),
                "gjx_logpdf_grad_f32")
    return outs


def axpy(a: float, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """y + a * x over the broadcast shape (multiply and add rounded separately)."""
    shp = _bshape(_f32(x), y)
    xe, ye = x.expand(shp).contiguous(), _f32(y).expand(shp).contiguous()
    out = torch.empty_like(xe)
    if xe.numel() > 0:
        N.check(N.lib().gjx_axpy_f32(float(a), _ptr(xe), 1, _ptr(ye), 1, xe.numel(), _ptr(out), _stream()),
                "gjx_axpy_f32")
    return out


def sum_diff(a, b, c=None, d=None) -> torch.Tensor:
    """(a + b) - (c + d) elementwise; (a + b) - c when d is None; (a + b) when c, d are None."""
    ts = [t for t in (a, b, c, d) if t is not None]
    shp = tuple(torch.broadcast_shapes(*[tuple(t.shape) for t in ts]))
    e = [None if t is None else _f32(t).expand(shp).contiguous() for t in (a, b, c, d)]
    out = torch.empty(shp, dtype=torch.float32, device=e[0].device)
    if out.numel() > 0:
        N.check(N.lib().gjx_sum_diff_f32(_ptr(e[0]), _ptr(e[1]), _ptr(e[2]), _ptr(e[3]), out.numel(), _ptr(out),
                                         _stream()), "gjx_sum_diff_f32")
    return out


def sub_scalar(a: torch.Tensor, d_scalar: torch.Tensor) -> torch.Tensor:
    a = _f32(a).contiguous()
    out = torch.empty_like(a)
    if a.numel() > 0:
        N.check(N.lib().gjx_sub_scalar_f32(_ptr(a), _ptr(_f32(d_scalar)), a.numel(), _ptr(out), _stream()),
                "gjx_sub_scalar_f32")
    return out


def weighted_sum(logw: torch.Tensor, d_lse: torch.Tensor, values: torch.Tensor) -> torch.Tensor:
    """sum_i exp(logw[i] - lse) values[i, ...] -> tensor of the trailing shape of `values`."""
    logw = _f32(logw).contiguous()
    n = logw.numel()
    v = _f32(values).contiguous()
    if v.shape[0] != n:
        raise GjxError("weighted_sum: the leading axis of `values` must be the particle axis")
    inner = max(1, v.numel() // n)
    out = torch.empty(inner, dtype=torch.float32, device=logw.device)
    N.check(N.lib().gjx_weighted_sum_f32(_ctx_for(logw).handle, _ptr(logw), _ptr(_f32(d_lse)), _ptr(v), n, inner,
                                         _ptr(out), _stream()), "gjx_weighted_sum_f32")
    return out.reshape(v.shape[1:])


def categorical_logpdf(x, logits):
    """x: (n,) int32 or one shared value; logits: (n, k) or (k,).  Result: one entry per row of the broadcast."""
    logits = _f32(logits).contiguous()
    x = torch.as_tensor(x, device=logits.device).to(torch.int32)
    if logits.dim() > 2 or x.dim() > 1:
        raise N.GjxError(f"categorical_logpdf: x of shape {tuple(x.shape)} against logits {tuple(logits.shape)}")
    k = logits.shape[-1]
    stride = 1 if logits.dim() == 2 else 0
    if stride and x.numel() != logits.shape[0]:
        if x.numel() != 1:
            raise N.GjxError(f"categorical_logpdf: {x.numel()} values against {logits.shape[0]} rows of logits")
        x = x.reshape(1).expand(logits.shape[0])          # a constraint shared by every particle
    scalar = x.dim() == 0
    x = x.reshape(-1).contiguous()
    out = torch.empty(x.numel(), dtype=torch.float32, device=x.device)
    if x.numel() == 0:
        return out
    N.check(N.lib().gjx_categorical_logpdf_f32(_ptr(x), _ptr(logits), stride, x.numel(), k, _ptr(out),
                                               _stream()), "gjx_categorical_logpdf_f32")
    return out.reshape(()) if scalar else out


def mvnormal_logpdf(x, mu, cov):
    """x: (n, d); mu: (n, d) or (d,); cov: (n, d, d) or (d, d)."""
    x = _f32(x)
    mu = _f32(mu).contiguous()
    cov = _f32(cov).contiguous()
    rows = mu.shape[0] if mu.dim() == 2 else (cov.shape[0] if cov.dim() == 3 else x.shape[0])
    if x.shape[0] != rows:
        if x.shape[0] != 1:
            raise N.GjxError(f"mvnormal_logpdf: {x.shape[0]} values against {rows} rows of parameters")
        x = x.expand(rows, x.shape[1])                      # an observation shared by every particle
    x = x.contiguous()
    n, d = x.shape
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    if n == 0:
        return out
    N.check(N.lib().gjx_mvnormal_logpdf_f32(_ptr(x), _ptr(mu), 1 if mu.dim() == 2 else 0, _ptr(cov),
                                            1 if cov.dim() == 3 else 0, n, d, _ptr(out), _stream()),
            "gjx_mvnormal_logpdf_f32")
    return out


# --------------------------------------------------------------------------- #
# MH accept / select for generic traces
# --------------------------------------------------------------------------- #
def mh_accept(k, log_weight: torch.Tensor, n: int, offset: int = 0) -> torch.Tensor:
    """accept (uint8, n) = log(uniform(k)) < minimum(0, log_weight)  (mcmc.py:293-296)."""
    lw = _f32(log_weight).contiguous().reshape(-1)
    stride = 1 if lw.numel() == n and n > 1 else (1 if n == 1 else 0)
    out = torch.empty(n, dtype=torch.uint8, device=lw.device)
    N.check(N.lib().gjx_mh_accept(_ckey(k), offset, n, _ptr(lw), stride, _ptr(out), _stream()), "gjx_mh_accept")
    return out


def select_rows(accept: torch.Tensor, new: torch.Tensor, old: torch.Tensor) -> torch.Tensor:
    """where(accept[:, None, ...], new, old) for 4-byte leaves whose leading axis is the particle axis."""
    n = accept.numel()
    if new.dtype != old.dtype or new.element_size() != 4 or new.shape != old.shape or new.shape[0] != n:
        raise GjxError("select_rows: leaves must be 4-byte tensors of equal shape with the particle axis first")
    a, b = new.contiguous(), old.contiguous()
    out = torch.empty_like(a)
    if a.numel() == 0:
        return out
    N.check(N.lib().gjx_select_rows_b32(_ptr(accept), _ptr(a), _ptr(b), n, a.numel() // n, _ptr(out), _stream()),
            "gjx_select_rows_b32")
    return out


# --------------------------------------------------------------------------- #
# normalisation / resampling
# --------------------------------------------------------------------------- #
def logsumexp_ess(logw: torch.Tensor) -> torch.Tensor:
    """Device tensor {max, logsumexp, ess, sum exp, sum exp^2} (no host sync)."""
    logw = _f32(logw).contiguous()
    if logw.numel() == 0:       # the empty collection: logsumexp of nothing is -inf, its ESS undefined
        return torch.tensor([-math.inf, -math.inf, math.nan, 0.0, 0.0], dtype=torch.float32, device=logw.device)
    out = torch.empty(5, dtype=torch.float32, device=logw.device)
    ctx = _ctx_for(logw)
    N.check(N.lib().gjx_logsumexp_ess(ctx.handle, _ptr(logw), logw.numel(), _ptr(out), _stream()),
            "gjx_logsumexp_ess")
    return out


def max_f32(a: torch.Tensor) -> torch.Tensor:
    a = _f32(a).contiguous()
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    N.check(N.lib().gjx_max_f32(_ctx_for(a).handle, _ptr(a), a.numel(), _ptr(out), _stream()), "gjx_max_f32")
    return out


def sumexp_f32(a: torch.Tensor, d_max: torch.Tensor) -> torch.Tensor:
    a = _f32(a).contiguous()
    out = torch.empty(2, dtype=torch.float32, device=a.device)
    N.check(N.lib().gjx_sumexp_f32(_ctx_for(a).handle, _ptr(a), _ptr(d_max), a.numel(), _ptr(out), _stream()),
            "gjx_sumexp_f32")
    return out


def normalized_weights(logw: torch.Tensor, d_lse: torch.Tensor) -> torch.Tensor:
    logw = _f32(logw).contiguous()
    out = torch.empty_like(logw)
    N.check(N.lib().gjx_normalized_weights(_ptr(logw), _ptr(d_lse), logw.numel(), _ptr(out), _stream()),
            "gjx_normalized_weights")
    return out


def cumsum(w: torch.Tensor) -> torch.Tensor:
    """jnp.cumsum(w) with lax.associative_scan's f32 association."""
    w = _f32(w).contiguous()
    out = torch.empty_like(w)
    N.check(N.lib().gjx_cumsum_f32(_ctx_for(w).handle, _ptr(w), w.numel(), _ptr(out), _stream()),
            "gjx_cumsum_f32")
    return out


def cumsum_tile_totals(w: torch.Tensor) -> torch.Tensor:
    w = _f32(w).contiguous()
    nt = (w.numel() + N.SCAN_TILE - 1) // N.SCAN_TILE
    out = torch.empty(nt, dtype=torch.float32, device=w.device)
    N.check(N.lib().gjx_cumsum_tile_totals(_ptr(w), w.numel(), _ptr(out), _stream()), "gjx_cumsum_tile_totals")
    return out


def cumsum_apply(w: torch.Tensor, scanned_totals: torch.Tensor, tile_offset: int) -> torch.Tensor:
    w = _f32(w).contiguous()
    out = torch.empty_like(w)
    N.check(N.lib().gjx_cumsum_apply(_ptr(w), w.numel(), _ptr(scanned_totals.contiguous()), tile_offset,
                                     _ptr(out), _stream()), "gjx_cumsum_apply")
    return out


def searchsorted_systematic(cdf: torch.Tensor, u: float, j0: int, m: int, n_total: int) -> torch.Tensor:
    cdf = _f32(cdf).contiguous()
    out = torch.empty(m, dtype=torch.int32, device=cdf.device)
    N.check(N.lib().gjx_searchsorted_systematic(_ptr(cdf), cdf.numel(), float(u), j0, m, n_total, _ptr(out),
                                                _stream()), "gjx_searchsorted_systematic")
    return out


_plan_cache: dict = {}


def bisect_levels(n_cdf: int) -> int:
    return int(N.lib().gjx_bisect_levels(int(n_cdf)))


def bisect_plan(n_cdf: int, lc: int) -> np.ndarray:
    """Probe index (mid) of every node of the first lc bisection levels, heap order (host)."""
    key_ = (int(n_cdf), int(lc))
    if key_ not in _plan_cache:
        out = np.zeros(1 << lc, np.int32)
        N.check(N.lib().gjx_bisect_plan(int(n_cdf), int(lc), out.ctypes.data_as(C.POINTER(C.c_int32))),
                "gjx_bisect_plan")
        _plan_cache[key_] = out
    return _plan_cache[key_]


def searchsorted_window(coarse: torch.Tensor, lc: int, n_cdf: int, u: float, j0: int, m: int,
                        n_total: int, n_ranges: int = 1) -> torch.Tensor:
    """Device int64 (n_ranges, 2): {low of the first position, high of the last} after lc coarse
    levels, for the consecutive position ranges [j0 + r m, j0 + (r + 1) m)."""
    out = torch.empty((n_ranges, 2), dtype=torch.int64, device=coarse.device)
    N.check(N.lib().gjx_searchsorted_window(_ptr(_f32(coarse).contiguous()), lc, n_cdf, float(u), j0, m, n_ranges,
                                            n_total, _ptr(out), _stream()), "gjx_searchsorted_window")
    return out


def searchsorted_systematic_2l(coarse: torch.Tensor, lc: int, cdf_slab: torch.Tensor, slab_base: int,
                               n_cdf: int, u: float, j0: int, m: int, n_total: int) -> torch.Tensor:
    cdf_slab = _f32(cdf_slab).contiguous()
    out = torch.empty(m, dtype=torch.int32, device=cdf_slab.device)
    N.check(N.lib().gjx_searchsorted_systematic_2l(_ptr(_f32(coarse).contiguous()), lc, _ptr(cdf_slab), slab_base,
                                                   cdf_slab.numel(), n_cdf, float(u), j0, m, n_total, _ptr(out),
                                                   _stream()), "gjx_searchsorted_systematic_2l")
    return out


def resample_systematic(weights: torch.Tensor, u: float, return_cdf: bool = False):
    """Ancestors (int32) of systematic_resample given normalised weights and u."""
    w = _f32(weights).contiguous()
    anc = torch.empty(w.numel(), dtype=torch.int32, device=w.device)
    cdf = torch.empty_like(w) if return_cdf else None
    N.check(N.lib().gjx_resample_systematic(_ctx_for(w).handle, _ptr(w), w.numel(), float(u), _ptr(anc),
                                            _ptr(cdf), _stream()), "gjx_resample_systematic")
    return (anc, cdf) if return_cdf else anc


def cumsum_normalized(logw: torch.Tensor, d_lse: torch.Tensor) -> torch.Tensor:
    """jnp.cumsum(exp(logw - lse)) with the normalisation fused into the scan's loads."""
    lw = _f32(logw).contiguous()
    out = torch.empty_like(lw)
    N.check(N.lib().gjx_cumsum_normalized_f32(_ctx_for(lw).handle, _ptr(lw), _ptr(_f32(d_lse)), lw.numel(), _ptr(out),
                                              _stream()), "gjx_cumsum_normalized_f32")
    return out


def resample_systematic_logw(logw: torch.Tensor, d_lse: torch.Tensor, u: float, return_cdf: bool = False):
    """Ancestors (int32) of systematic_resample straight from the log-weights and their lse."""
    lw = _f32(logw).contiguous()
    anc = torch.empty(lw.numel(), dtype=torch.int32, device=lw.device)
    cdf = torch.empty_like(lw) if return_cdf else None
    N.check(N.lib().gjx_resample_systematic_logw(_ctx_for(lw).handle, _ptr(lw), _ptr(_f32(d_lse)), lw.numel(),
                                                 float(u), _ptr(anc), _ptr(cdf), _stream()), "gjx_resample_systematic_logw")
    return (anc, cdf) if return_cdf else anc


STATIC_MAX_N = 1 << 22   # GJX_STATIC_MAX_N (include/gjx.h)


def resample_systematic_logw_static(logw: torch.Tensor, d_lse: torch.Tensor, u: float,
                                    out: torch.Tensor | None = None) -> torch.Tensor:
    """gjx_resample_systematic_logw_static: the same ancestors from the library's static workspace;
    the C call allocates nothing (pass `out`, n int32, so that this wrapper does not either)."""
    lw = _f32(logw)
    if not lw.is_contiguous():
        raise GjxError("resample_systematic_logw_static: log-weights must be contiguous")
    if lw.device.index != torch.cuda.current_device():
        raise GjxError("resample_systematic_logw_static: the tensors' device must be the current device")
    anc = torch.empty(lw.numel(), dtype=torch.int32, device=lw.device) if out is None else out
    if anc.dtype != torch.int32 or anc.numel() != lw.numel():
        raise GjxError("resample_systematic_logw_static: `out` must hold n int32")
    if not anc.is_contiguous() or anc.device != lw.device:
        raise GjxError("resample_systematic_logw_static: `out` must be contiguous and on the log-weights' device")
    lse = _f32(d_lse)
    if lse.device != lw.device or lse.numel() < 1:
        raise GjxError("resample_systematic_logw_static: the log-sum-exp scalar must live on the log-weights' device")
    N.check(N.lib().gjx_resample_systematic_logw_static(_ptr(lw), _ptr(lse), lw.numel(), float(u), _ptr(anc),
                                                        _stream()), "gjx_resample_systematic_logw_static")
    return anc


def resample_multinomial(k, logw: torch.Tensor, m: int | None = None) -> torch.Tensor:
    logw = _f32(logw).contiguous()
    m = logw.numel() if m is None else int(m)
    anc = torch.empty(m, dtype=torch.int32, device=logw.device)
    N.check(N.lib().gjx_resample_multinomial(_ckey(k), _ptr(logw), logw.numel(), m, _ptr(anc), _stream()),
            "gjx_resample_multinomial")
    return anc


def gather_columns(idx: torch.Tensor, cols: Sequence[torch.Tensor], idx_base: int = 0,
                   out: Sequence[torch.Tensor] | None = None) -> list[torch.Tensor]:
    """[col[idx - idx_base] for col in cols] in one launch per 6 columns."""
    m = idx.numel()
    cols = [_f32(c).contiguous() for c in cols]
    if out is None:
        out = [torch.empty(m, dtype=torch.float32, device=idx.device) for _ in cols]
    nc = len(cols)
    src = (C.c_void_p * nc)(*[c.data_ptr() for c in cols])
    dst = (C.c_void_p * nc)(*[o.data_ptr() for o in out])
    N.check(N.lib().gjx_gather_columns_f32(_ptr(idx), idx_base, m, src, dst, nc, _stream()),
            "gjx_gather_columns_f32")
    return list(out)


# --------------------------------------------------------------------------- #
# localization family
# --------------------------------------------------------------------------- #
def make_loc_world(width, height, walls, n_rays=8, max_range=10.0, sensor_sigma=0.3,
                   init_mean=(1.5, 1.5, 0.0), init_sigma=(0.5, 0.5, 0.3),
                   drift_sigma=(0.25, 0.25, 0.1), clip_margin=0.1) -> LocWorld:
    walls = np.asarray(walls, dtype=np.float32).reshape(-1, 4)
    if walls.shape[0] < 1:
        raise ValueError(f"World must have at least 1 internal wall, got {walls.shape[0]}")
    if walls.shape[0] > N.MAX_WALLS:
        raise GjxError(f"n_walls={walls.shape[0]} exceeds GJX_MAX_WALLS={N.MAX_WALLS}")
    if not (1 <= n_rays <= N.MAX_RAYS):
        raise GjxError(f"n_rays={n_rays} not in [1, {N.MAX_RAYS}]")
    w = LocWorld()
    w.width, w.height = float(width), float(height)
    w.n_walls, w.n_rays = walls.shape[0], int(n_rays)
    for i, row in enumerate(walls):
        for j in range(4):
            w.walls[i][j] = float(row[j])
    w.max_range, w.sensor_sigma, w.clip_margin = float(max_range), float(sensor_sigma), float(clip_margin)
    for i in range(3):
        w.init_mean[i] = float(init_mean[i])
        w.init_sigma[i] = float(init_sigma[i])
        w.drift_sigma[i] = float(drift_sigma[i])
    return w


def _obs(obs, R):
    o = np.ascontiguousarray(np.asarray(obs, dtype=np.float32).reshape(-1))
    if o.size != R:
        raise GjxError(f"observation has {o.size} rays, world has {R}")
    return o.ctypes.data_as(C.POINTER(C.c_float)), o


def _keys_block(keys, shape):
    k = np.ascontiguousarray(np.asarray(keys, dtype=np.uint32).reshape(shape))
    return k.ctypes.data_as(C.POINTER(C.c_uint32)), k


def loc_lidar(world: LocWorld, px, py, theta) -> torch.Tensor:
    n = px.numel()
    out = torch.empty((n, world.n_rays), dtype=torch.float32, device=px.device)
    N.check(N.lib().gjx_loc_lidar(C.byref(world), _ptr(px), _ptr(py), _ptr(theta), n, _ptr(out), _stream()),
            "gjx_loc_lidar")
    return out


def loc_lidar_jac(world: LocWorld, px, py, theta):
    """-> (distances (n, n_rays), jacobian (3, n, n_rays): d/dpx, d/dpy, d/dtheta of every distance)."""
    n = px.numel()
    out = torch.empty((n, world.n_rays), dtype=torch.float32, device=px.device)
    jac = torch.empty((3, n, world.n_rays), dtype=torch.float32, device=px.device)
    N.check(N.lib().gjx_loc_lidar_jac(C.byref(world), _ptr(px), _ptr(py), _ptr(theta), n, _ptr(out), _ptr(jac), _stream()),
            "gjx_loc_lidar_jac")
    return out, jac


def loc_ray_scores(world: LocWorld, obs, x, y, th) -> torch.Tensor:
    """(n_rays, n) per-ray measurement scores minus the constant: 0.5 ((obs_r - d_r) / sigma)^2."""
    n = x.numel()
    po, _o = _obs(obs, world.n_rays)
    out = torch.empty((world.n_rays, n), dtype=torch.float32, device=x.device)
    N.check(N.lib().gjx_loc_ray_scores(C.byref(world), po, _ptr(_f32(x).contiguous()), _ptr(_f32(y).contiguous()),
                                       _ptr(_f32(th).contiguous()), n, _ptr(out), _stream()), "gjx_loc_ray_scores")
    return out


def loc_lidar_host(world: LocWorld, px, py, theta) -> np.ndarray:
    """Host (CPU, no GPU needed) lidar used by the synthetic-data generator."""
    px, py, theta = (np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(-1)) for v in (px, py, theta))
    out = np.zeros((px.size, world.n_rays), np.float32)
    fp = C.POINTER(C.c_float)
    N.check(N.lib().gjx_loc_lidar_host(C.byref(world), px.ctypes.data_as(fp), py.ctypes.data_as(fp),
                                       theta.ctypes.data_as(fp), px.size, out.ctypes.data_as(fp)),
            "gjx_loc_lidar_host")
    return out


def host_normal(k) -> np.float32:
    """jax.random.normal(k) (scalar) computed on the host."""
    return np.float32(N.lib().gjx_host_normal(_ckey(k)))


def loc_extend(world: LocWorld, keys3, is_initial: bool, obs, prev, logw_in, out=None, offset: int = 0,
               ray_scores: torch.Tensor | None = None):
    """gjx_loc_extend.  prev = (px, py, pth) previous pose columns (None when is_initial);
    out = dict of 7 output tensors x,y,th,mx,my,mth,logw (allocated when None; may alias the
    inputs for an in-place update).  ray_scores: optional (n_rays, n) tensor that receives the
    per-ray measurement scores (gjx_loc_extend_scores).  Returns the dict."""
    n = logw_in.numel()
    dev = logw_in.device
    if out is None:
        out = {k_: torch.empty(n, dtype=torch.float32, device=dev)
               for k_ in ("x", "y", "th", "mx", "my", "mth", "logw")}
    po, _o = _obs(obs, world.n_rays)
    pk, _k = _keys_block(keys3, (3, 2))
    px, py, pth = prev if prev is not None else (None, None, None)
    if ray_scores is not None and tuple(ray_scores.shape) != (world.n_rays, n):
        raise GjxError(f"ray_scores must be ({world.n_rays}, {n}), got {tuple(ray_scores.shape)}")
    N.check(N.lib().gjx_loc_extend_scores(_ctx_for(logw_in).handle, C.byref(world), pk, int(bool(is_initial)), offset,
                                          n, po, _ptr(px), _ptr(py), _ptr(pth), _ptr(logw_in), _ptr(out["x"]),
                                          _ptr(out["y"]), _ptr(out["th"]), _ptr(out["mx"]), _ptr(out["my"]),
                                          _ptr(out["mth"]), _ptr(out["logw"]), *_rows(ray_scores), _stream()),
            "gjx_loc_extend_scores")
    return out


def loc_extend_locopt(world: LocWorld, keys3, grid_size: int, noise_std: float, grid_radius: float, obs, prev,
                      logw_in, out=None, offset: int = 0):
    """gjx_loc_extend_locopt: extend() with the locally-optimal grid proposal.  prev = the previous
    CHOICE columns (x, y, theta)."""
    n = logw_in.numel()
    if out is None:
        out = {k_: torch.empty(n, dtype=torch.float32, device=logw_in.device)
               for k_ in ("x", "y", "th", "mx", "my", "mth", "logw")}
    po, _o = _obs(obs, world.n_rays)
    pk, _k = _keys_block(keys3, (3, 2))
    N.check(N.lib().gjx_loc_extend_locopt(_ctx_for(logw_in).handle, C.byref(world), pk, int(grid_size),
                                          float(noise_std), float(grid_radius), offset, n, po, _ptr(prev[0]),
                                          _ptr(prev[1]), _ptr(prev[2]), _ptr(logw_in), _ptr(out["x"]),
                                          _ptr(out["y"]), _ptr(out["th"]), _ptr(out["mx"]), _ptr(out["my"]),
                                          _ptr(out["mth"]), _ptr(out["logw"]), _stream()), "gjx_loc_extend_locopt")
    return out


def loc_move_keys(scan_key, n_moves: int, kind: str, sel_mask: int = 7) -> np.ndarray:
    """Host key schedule of the K rejuvenation moves -> (n_moves, 4, 2) uint32."""
    out = np.zeros((max(n_moves, 0), 4, 2), np.uint32)
    if n_moves > 0:
        N.check(N.lib().gjx_loc_move_keys(_ckey(scan_key), n_moves, {"mh": 1, "hmc": 2}[kind], int(sel_mask),
                                          out.ctypes.data_as(C.POINTER(C.c_uint32))), "gjx_loc_move_keys")
    return out


def new_event():
    h = C.c_void_p()
    N.check(N.lib().gjx_event_create(C.byref(h)), "gjx_event_create")
    return h


def event_destroy(ev) -> None:
    N.check(N.lib().gjx_event_destroy(ev), "gjx_event_destroy")


def spin_us(microseconds: int) -> None:
    """Keeps the current stream busy for about `microseconds` with one idle wave (gjx_spin_us)."""
    N.check(N.lib().gjx_spin_us(int(microseconds), _stream()), "gjx_spin_us")


N_STAGE_EVENTS = 7
STAGES = ("extend", "normalise", "host_decision", "scan", "search", "moves")   # the six intervals between them


def set_stage_events(ctx: "Context", events) -> None:
    """Stage timing of the fused smc-step entry points on `ctx`: 7 events from new_event(), or None to end it."""
    if events is None:
        N.check(N.lib().gjx_ctx_set_stage_events(ctx.handle, None, 0), "gjx_ctx_set_stage_events")
        return
    arr = (C.c_void_p * N_STAGE_EVENTS)(*[e.value for e in events])
    N.check(N.lib().gjx_ctx_set_stage_events(ctx.handle, arr, N_STAGE_EVENTS), "gjx_ctx_set_stage_events")


def event_elapsed_ms(begin, end) -> float:
    ms = C.c_float()
    N.check(N.lib().gjx_event_elapsed_ms(begin, end, C.byref(ms)), "gjx_event_elapsed_ms")
    return float(ms.value)


class LocStepBuffers:
    """Persistent device buffers of the fused smc-step entry points: two column sets
    {x, y, th, mx, my, mth} (ping-pong across resampling), log-weights, statistics."""

    def __init__(self, n: int, device, n_rays: int = 0, halo: int = 0):
        # every column is the head of an (n + halo)-element row: the sharded step writes the rows it receives
        # from other shards behind its own n particles, so that ancestors outside the own shard are plain
        # indices into the same column (halo = 0 on one GPU)
        self.halo, self.pitch = int(halo), n + int(halo)
        ext = lambda rows: torch.empty((rows, self.pitch), dtype=torch.float32, device=device)  # noqa: E731
        self._ext_sets = [ext(6), ext(6)]
        self.sets = [[e[c, :n] for c in range(6)] for e in self._ext_sets]
        # per-ray measurement scores of the particles as extended (the trace's stored old_score):
        # written by loc_smc_extend, read by the mh launch of the same step (n_rays = 0: recompute)
        self._ext_scores = ext(n_rays) if n_rays > 0 else None
        self.ray_scores = self._ext_scores[:, :n] if n_rays > 0 else None
        self.ptrs = [(C.c_void_p * 6)(*[t.data_ptr() for t in s_]) for s_ in self.sets]
        self.cur = 0
        self.logw = torch.zeros(n, dtype=torch.float32, device=device)
        self.stats = torch.empty(5, dtype=torch.float32, device=device)
        self.n = n
        self.ctx = Context.get(torch.device(device).index or 0, n)
        # the step with the decision on the device (loc_smc_step_auto): {1, n, take, flags}, the search range,
        # the device's running log-ML estimate, {ESS, take} of the last step
        self.decision = torch.tensor([1, n, 0, 0], dtype=torch.int32, device=device)
        self.range2 = torch.zeros(2, dtype=torch.int64, device=device)
        self.dev_lml = torch.zeros(1, dtype=torch.float32, device=device)
        self.trace2 = torch.zeros(2, dtype=torch.float32, device=device)

    @property
    def cols(self):
        return self.sets[self.cur]


def _scores_for(buf: LocStepBuffers, world: LocWorld):
    rs = buf.ray_scores
    if rs is not None and rs.shape[0] != world.n_rays:
        raise GjxError(f"step buffers hold scores for {rs.shape[0]} rays, the world has {world.n_rays}")
    return rs


def loc_smc_extend(world: LocWorld, buf: LocStepBuffers, keys3, is_initial: bool, obs, offset: int = 0):
    """extend (in place on the current column set) + normaliser -> buf.stats (device)."""
    po, _o = _obs(obs, world.n_rays)
    pk, _k = _keys_block(keys3, (3, 2))
    c = buf.cols
    prev = [C.c_void_p(0)] * 3 if is_initial else [C.c_void_p(t.data_ptr()) for t in c[:3]]
    N.check(N.lib().gjx_loc_smc_extend(buf.ctx.handle, C.byref(world), pk, int(bool(is_initial)), offset, buf.n, po,
                                       *prev, _ptr(buf.logw), *[C.c_void_p(t.data_ptr()) for t in c], _ptr(buf.logw),
                                       _ptr(buf.stats), *_rows(_scores_for(buf, world)), _stream()), "gjx_loc_smc_extend")
    return buf.stats


def loc_smc_presearch(buf: LocStepBuffers, u: float) -> None:
    """The scan and the search of the resampling this step may take, enqueued before the ESS is read (the ancestors
    stay in the context; loc_smc_resample_moves(..., presearched=True) uses them)."""
    N.check(N.lib().gjx_loc_smc_presearch(buf.ctx.handle, _ptr(buf.logw), _ptr(buf.stats), buf.n, float(u), _stream()),
            "gjx_loc_smc_presearch")


def loc_smc_resample_moves(world: LocWorld, buf: LocStepBuffers, do_resample: bool, u: float, kind: str,
                           move_keys, is_initial: bool, obs, step_size: float = 0.0, n_steps: int = 1,
                           sel_mask: int = 7, offset: int = 0, accept_counts: torch.Tensor | None = None,
                           events=None, use_scores: bool = True, presearched: bool = False):
    """Optional systematic resampling + fused moves; flips buf.cur when it resampled.
    presearched: loc_smc_presearch already ran for this step (its ancestors are used when do_resample, ignored else).
    events: optional (begin, end) pair from new_event(), recorded around the move kernel.
    use_scores: buf.ray_scores (when allocated) hold the scores written by loc_smc_extend for the
    particles in buf.cols and this observation; pass False when that is not the case (e.g. the
    particles were exchanged between shards since)."""
    po, _o = _obs(obs, world.n_rays)
    mk = np.ascontiguousarray(np.asarray(move_keys, np.uint32).reshape(-1, 4, 2))
    if mk.shape[0] > N.MAX_MOVES:
        raise GjxError(f"at most {N.MAX_MOVES} fused moves per call")
    k = {"none": 0, "mh": 1, "hmc": 2}[kind] if mk.shape[0] > 0 else 0
    N.check(N.lib().gjx_loc_smc_resample_moves(
        buf.ctx.handle, C.byref(world), (2 if presearched else 1) if do_resample else 0, float(u), _ptr(buf.stats), k,
        mk.ctypes.data_as(C.POINTER(C.c_uint32)), mk.shape[0], int(sel_mask), float(step_size), int(n_steps),
        int(bool(is_initial)), offset, buf.n, po, buf.ptrs[buf.cur], buf.ptrs[1 - buf.cur], _ptr(buf.logw),
        *_rows(_scores_for(buf, world) if use_scores else None),
        _ptr(accept_counts), events[0] if events else None, events[1] if events else None, _stream()),
        "gjx_loc_smc_resample_moves")
    if do_resample:
        buf.cur = 1 - buf.cur


def loc_smc_step_auto(world: LocWorld, buf: LocStepBuffers, keys3, is_initial: bool, obs, u: float, move_keys,
                      sel_mask: int = 7, accept_counts: torch.Tensor | None = None, events=None, ev_stats=None):
    """One whole mh step in one call, the ESS decision taken on the device (buf.decision[2]); always flips buf.cur.
    ev_stats: raw event handle recorded once buf.stats is final (fetch it on another stream meanwhile)."""
    po, _o = _obs(obs, world.n_rays)
    pk, _k = _keys_block(keys3, (3, 2))
    mk = _move_key_block(move_keys)
    N.check(N.lib().gjx_loc_smc_step_auto(
        buf.ctx.handle, C.byref(world), pk, int(bool(is_initial)), buf.n, po, float(u),
        mk.ctypes.data_as(C.POINTER(C.c_uint32)), mk.shape[0], int(sel_mask), buf.ptrs[buf.cur], buf.ptrs[1 - buf.cur],
        _ptr(buf.logw), _ptr(buf.stats), *_rows(_scores_for(buf, world)), _ptr(buf.decision), _ptr(buf.range2),
        _ptr(buf.dev_lml), _ptr(buf.trace2), _ptr(accept_counts), ev_stats, events[0] if events else None,
        events[1] if events else None, _stream()), "gjx_loc_smc_step_auto")
    buf.cur = 1 - buf.cur


# --------------------------------------------------------------------------- #
# sharded smc step (one process per GPU; collectives run in genjax_amd.parallel)
# --------------------------------------------------------------------------- #
SHARD_ROW = 7          # floats of an exchanged row: {first position (int32 bits), x, y, th, mx, my, mth}
FINE_LEVELS = 11       # bisection levels left after the coarse table: intervals of <= 2^11 entries
SCAN_TILE = 2048
MAX_RANKS = 64


PROBE_REC = 12         # floats of a coarse-table probe record in the CDF message (gjx_shard_cdf_begin)


def shard_probe_lists(n_global: int, lc: int, world: int):
    """The probes (nodes 1 .. 2^lc - 1 of the bisection tree's first lc levels) grouped by the rank whose shard
    holds the probe's mid: (nodes, mids), int32 (world, p_cap), node 0 = empty slot; p_cap >= 1."""
    n = n_global // world
    mids = np.asarray(bisect_plan(n_global, lc)).astype(np.int64)
    ids = np.arange(1, 1 << lc, dtype=np.int64)
    owner = mids[1:] // n
    per = [ids[owner == r] for r in range(world)]
    p_cap = max(1, max(len(p_) for p_ in per))
    nodes = np.zeros((world, p_cap), np.int32)
    m_out = np.zeros((world, p_cap), np.int32)
    for r, p_ in enumerate(per):
        nodes[r, :len(p_)] = p_
        m_out[r, :len(p_)] = mids[p_]
    return nodes, m_out


class ShardStepBuffers(LocStepBuffers):
    """LocStepBuffers plus the persistent workspace of the sharded step of rank `rank`.  Columns (and score
    rows) carry a halo of n elements behind the own n particles: at most one received row per own position."""

    def __init__(self, n: int, n_global: int, world: int, rank: int, device, n_rays: int = 0):
        if n % SCAN_TILE != 0 or n * world != n_global or not (1 <= world <= MAX_RANKS) or n_global >= 1 << 31:
            raise GjxError(f"sharded step: n_local={n} must be a multiple of {SCAN_TILE}, n_global = n_local * world "
                           f"< 2^31, world <= {MAX_RANKS}")
        super().__init__(n, device, n_rays=n_rays, halo=n)
        f = lambda k, dt=torch.float32: torch.empty(k, dtype=dt, device=device)  # noqa: E731
        self.n_global, self.world, self.rank, self.offset = n_global, world, rank, rank * n
        nt = n // SCAN_TILE
        # normaliser: local max -> all_gather; tile sums -> all_gather; fold
        # (max1 holds -inf between steps: the one-call front end raises it by atomics and puts it back)
        self.max1, self.max_all = torch.full((1,), -math.inf, dtype=torch.float32, device=device), f(world)
        self.tiles, self.tiles_all = f(2 * nt), f(2 * nt * world)
        # what the host reads back per step, in one buffer (one copy): the five global statistics, the split of the
        # overlapped step {p0 + 1, p1, go, 0}, the row counts [source, destination]
        self.report = torch.zeros(16 + world * world, dtype=torch.int32, device=device)
        self.gstats = self.report[:5].view(torch.float32)
        self.split, self.counts_all = self.report[8:12], self.report[16:]
        # CDF: message {tile totals, first tile, last tile, one record per owned coarse-table probe} -> all_gather;
        # every rank then builds the complete coarse table itself (no all_reduce)
        self.lc = max(0, bisect_levels(n_global) - FINE_LEVELS)
        nodes, mids = shard_probe_lists(n_global, self.lc, world)
        self.p_cap = nodes.shape[1]
        self.probe_nodes = torch.from_numpy(nodes).to(device)          # (world, p_cap), 0 = empty slot
        self.probe_mids = torch.from_numpy(mids).to(device)
        msg_len = nt + 2 * SCAN_TILE + PROBE_REC * self.p_cap
        self.msg, self.msg_all = f(msg_len), f(msg_len * world)
        self.all_totals, self.scanned = f(nt * world), f(nt * world)
        self.cdf_ext = f(n + 2 * SCAN_TILE)
        self.cdf = self.cdf_ext[SCAN_TILE:SCAN_TILE + n]
        self.table = f(1 << self.lc)
        # owner-side search: any number of the n_global positions can select this shard
        self.cand_cap = n_global
        self.range = f(2, torch.int64)
        self.cand_idx = f(self.cand_cap, torch.int32)
        self.tile_counts, self.tile_offsets = f(self.cand_cap // SCAN_TILE + 1, torch.int32), f(self.cand_cap // SCAN_TILE + 1, torch.int32)
        self.counts = f(world, torch.int32)
        # rows: sent <= one per own particle and destination boundary (full capacity up front: the pack kernel is
        # enqueued before the host knows the count), received <= one per own position (grown on demand)
        self._send, self._recv = f(SHARD_ROW * (n + world)), f(SHARD_ROW * 4096)
        self.anc = f(n, torch.int32)
        self.err = torch.zeros(1, dtype=torch.int32, device=device)
        self._args = {}                                  # marshalled argument tuples of the per-step calls
        self._front = {}                                 # gjx_shard_bufs of either column set (shard_front)
        self.anc_fix = f(n, torch.int32)                # overlapped step: ancestors of the positions outside [p0, p1)

    def parse_report(self, rep: np.ndarray):
        """Host copy of self.report -> (gstats f32[5], counts int32[world^2], split int32[4])."""
        return rep[:5].view(np.float32), rep[16:], rep[8:12]

    def rows(self, name: str, rows: int) -> torch.Tensor:
        """The first `rows` rows of the send / receive buffer, grown geometrically up to its bound."""
        t = getattr(self, name)
        if t.numel() < rows * SHARD_ROW:
            t = torch.empty(max(rows * SHARD_ROW, 2 * t.numel()), dtype=torch.float32, device=t.device)
            setattr(self, name, t)
        return t[:rows * SHARD_ROW]


class CommHandle:
    """gjx_comm: the C side's communicator of the sharded step (RCCL bound at run time; one rank: none needed)."""

    def __init__(self, rank: int, world: int, unique_id: bytes | None = None):
        h = C.c_void_p()
        idp = None
        if world > 1 and unique_id is None:
            raise GjxError("CommHandle: more than one rank needs the 128-byte id of comm_unique_id()")
        if unique_id is not None:
            if len(unique_id) != 128:
                raise GjxError("CommHandle: the unique id has 128 bytes")
            idp = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        N.check(N.lib().gjx_comm_create(idp, int(rank), int(world), C.byref(h)), "gjx_comm_create")
        self.handle, self.rank, self.world = h, rank, world

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        if out.numel() * out.element_size() != self.world * inp.numel() * inp.element_size():
            raise GjxError("CommHandle.all_gather: out must hold world x inp")
        N.check(N.lib().gjx_comm_all_gather(self.handle, _ptr(inp), _ptr(out), inp.numel() * inp.element_size(),
                                            _stream()), "gjx_comm_all_gather")

    def close(self):
        if self.handle:
            N.lib().gjx_comm_destroy(self.handle)
            self.handle = None


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * 128)()
    N.check(N.lib().gjx_comm_unique_id(buf), "gjx_comm_unique_id")
    return bytes(buf)


def shard_front(world: LocWorld, buf: "ShardStepBuffers", comm: CommHandle, keys3, is_initial: bool, obs, u: float,
                move_keys, sel_mask: int = 7, accept_counts: torch.Tensor | None = None, ev_split=None, events=None):
    """The front end of one sharded mh step AND the move launch of its own-sourced positions in ONE call
    (gjx_shard_front): extend, the global normaliser, the CDF slice and coarse table, the owner-side search, the
    device's split / resampling decision, gjx_shard_moves_main -- with the three all_gathers issued from the C side.
    ev_split: raw event recorded between the split and the moves (the side work of the step forks there).
    Flips buf.cur (the moves write the alternate columns); buf.prev_cur names the set the side work still reads."""
    po, _o = _obs(obs, world.n_rays)
    pk, _k = _keys_block(keys3, (3, 2))
    mk = _move_key_block(move_keys)
    sb = buf._front.get(buf.cur)
    if sb is None:
        sb = N.ShardBufs()
        sb.n, sb.n_global, sb.rank, sb.n_ranks, sb.lc, sb.p_cap = buf.n, buf.n_global, buf.rank, buf.world, buf.lc, buf.p_cap
        for c in range(6):
            sb.cols[c] = buf.cols[c].data_ptr()
        sb.logw = buf.logw.data_ptr()
        rs = _scores_for(buf, world)
        sb.ray_scores = rs.data_ptr() if rs is not None else None
        sb.scores_stride = rs.stride(0) if rs is not None else 0
        for name in ("max1", "max_all", "tiles", "tiles_all", "gstats", "msg", "msg_all", "probe_nodes", "probe_mids",
                     "all_totals", "scanned", "cdf_ext", "table", "range", "cand_idx", "tile_counts", "counts", "split"):
            setattr(sb, name, getattr(buf, name).data_ptr())
        sb.msg_len, sb.cand_cap = buf.msg.numel(), buf.cand_cap
        buf._front[buf.cur] = sb
    N.check(N.lib().gjx_shard_front(buf.ctx.handle, comm.handle, C.byref(world), pk, int(bool(is_initial)), po, float(u),
                                    C.byref(sb), mk.shape[0], _ptr(accept_counts),
                                    mk.ctypes.data_as(C.POINTER(C.c_uint32)), int(sel_mask), buf.ptrs[1 - buf.cur], ev_split,
                                    events[0] if events else None, events[1] if events else None, _stream()),
            "gjx_shard_front")
    buf.cur = 1 - buf.cur


def new_shard_buffers(n: int, n_global: int, world: int, rank: int, device, n_rays: int = 0) -> ShardStepBuffers:
    return ShardStepBuffers(n, n_global, world, rank, device, n_rays=n_rays)


def loc_shard_extend(world: LocWorld, buf: ShardStepBuffers, keys3, is_initial: bool, obs):
    """extend (in place on the current column set) + this shard's maximum -> buf.max1 (device)."""
    po, _o = _obs(obs, world.n_rays)
    pk, _k = _keys_block(keys3, (3, 2))
    c = buf.cols
    prev = [C.c_void_p(0)] * 3 if is_initial else [C.c_void_p(t.data_ptr()) for t in c[:3]]
    N.check(N.lib().gjx_loc_shard_extend(buf.ctx.handle, C.byref(world), pk, int(bool(is_initial)), buf.offset, buf.n, po,
                                         *prev, _ptr(buf.logw), *[C.c_void_p(t.data_ptr()) for t in c], _ptr(buf.logw),
                                         _ptr(buf.max1), *_rows(_scores_for(buf, world)), _stream()),
            "gjx_loc_shard_extend")
    return buf.max1


def shard_local_max(buf: ShardStepBuffers) -> torch.Tensor:
    """Maximum of the current log-weights -> buf.max1 (outside a step: log_marginal_likelihood)."""
    N.check(N.lib().gjx_max_f32(buf.ctx.handle, _ptr(buf.logw), buf.n, _ptr(buf.max1), _stream()), "gjx_max_f32")
    return buf.max1


def _fixed_args(buf: ShardStepBuffers, name: str, build):
    """The arguments of a per-step call that only names the buffer's own persistent tensors, marshalled once."""
    a = buf._args.get(name)
    if a is None:
        a = buf._args[name] = build()
    return a


def shard_tile_sums(buf: ShardStepBuffers) -> torch.Tensor:
    """buf.max_all (every rank's maximum) -> buf.tiles: {sum exp, sum exp^2} of this shard's 2048-tiles."""
    a = _fixed_args(buf, "tile_sums", lambda: (_ptr(buf.logw), buf.n, _ptr(buf.max_all), buf.world, _ptr(buf.tiles)))
    N.check(N.lib().gjx_shard_tile_sums(*a, _stream()), "gjx_shard_tile_sums")
    return buf.tiles


def shard_stats_fold(buf: ShardStepBuffers) -> torch.Tensor:
    """buf.tiles_all (the tile sums of every shard) -> buf.gstats, the global five."""
    a = _fixed_args(buf, "stats_fold", lambda: (_ptr(buf.tiles_all), buf.tiles_all.numel() // 2, _ptr(buf.max_all),
                                                buf.world, _ptr(buf.gstats)))
    N.check(N.lib().gjx_shard_stats_fold(*a, _stream()), "gjx_shard_stats_fold")
    return buf.gstats


def shard_cdf_begin(buf: ShardStepBuffers) -> torch.Tensor:
    """buf.msg = tile totals of exp(logw - global lse), the raw first and last tile, one record per owned
    coarse-table probe."""
    a = _fixed_args(buf, "cdf_begin", lambda: (_ptr(buf.logw), C.c_void_p(buf.gstats.data_ptr() + 4), buf.n, buf.rank,
                                               _ptr(buf.probe_nodes[buf.rank]), _ptr(buf.probe_mids[buf.rank]),
                                               buf.p_cap, _ptr(buf.msg)))
    N.check(N.lib().gjx_shard_cdf_begin(*a, _stream()), "gjx_shard_cdf_begin")
    return buf.msg


def shard_cdf_finish(buf: ShardStepBuffers) -> torch.Tensor:
    """buf.msg_all -> this rank's slice of the global CDF with one tile of either neighbour (buf.cdf_ext)
    and the complete coarse bisection table (buf.table: every rank redoes the owners' additions)."""
    a = _fixed_args(buf, "cdf_finish", lambda: (
        buf.ctx.handle, _ptr(buf.logw), C.c_void_p(buf.gstats.data_ptr() + 4), buf.n, _ptr(buf.msg_all), buf.world,
        buf.rank, _ptr(buf.probe_nodes), _ptr(buf.probe_mids), buf.p_cap, _ptr(buf.all_totals), _ptr(buf.scanned),
        _ptr(buf.cdf_ext), buf.lc, _ptr(buf.table)))
    N.check(N.lib().gjx_shard_cdf_finish(*a, _stream()), "gjx_shard_cdf_finish")
    return buf.table


def shard_search(buf: ShardStepBuffers, u: float) -> torch.Tensor:
    """Ancestors of every position that can select this shard (buf.range, buf.cand_idx) and the rows this
    rank will send to every destination (buf.counts, int32 x world)."""
    a = _fixed_args(buf, "search", lambda: ((_ptr(buf.table), buf.lc, _ptr(buf.cdf_ext), buf.world, buf.rank, buf.n),
                                            (_ptr(buf.range), _ptr(buf.cand_idx), _ptr(buf.tile_counts),
                                             _ptr(buf.counts))))
    N.check(N.lib().gjx_shard_search(*a[0], float(u), buf.cand_cap, *a[1], _stream()), "gjx_shard_search")
    return buf.counts


def shard_count(buf: ShardStepBuffers) -> torch.Tensor:
    """The row counts of shard_search alone (buf.tile_counts, buf.counts): the one-call front end leaves them to the
    side work of the step."""
    a = _fixed_args(buf, "count", lambda: (_ptr(buf.cand_idx), _ptr(buf.range), buf.world, buf.rank, buf.n,
                                           _ptr(buf.tile_counts), _ptr(buf.counts)))
    N.check(N.lib().gjx_shard_count(*a, _stream()), "gjx_shard_count")
    return buf.counts


def shard_pack(buf: ShardStepBuffers, cols_set: int | None = None) -> torch.Tensor:
    """The sum(buf.counts) rows for the other ranks, grouped by destination ascending, at the head of the send
    buffer (whose capacity covers any count: the launch does not wait for the host to learn it).  cols_set: the
    column set that holds the particles as extended (default: the current one)."""
    a = _fixed_args(buf, "pack", lambda: (_ptr(buf.cand_idx), _ptr(buf.range), _ptr(buf.tile_counts),
                                          _ptr(buf.tile_offsets), buf.world, buf.rank, buf.n))
    N.check(N.lib().gjx_shard_pack(*a, buf.ptrs[buf.cur if cols_set is None else cols_set], _ptr(buf._send), _stream()),
            "gjx_shard_pack")
    return buf._send


def shard_send_rows(buf: ShardStepBuffers, n_rows: int) -> torch.Tensor:
    return buf._send[:n_rows * SHARD_ROW]


def shard_recv_rows(buf: ShardStepBuffers, rows: int) -> torch.Tensor:
    if rows > buf.halo:
        raise GjxError(f"sharded step: {rows} rows announced for a rank of {buf.n} positions")
    return buf.rows("_recv", rows)


def shard_resample_moves(world: LocWorld, buf: ShardStepBuffers, recv: torch.Tensor, kind: str, move_keys,
                         is_initial: bool, obs, step_size: float = 0.0, n_steps: int = 1, sel_mask: int = 7,
                         events=None, accept_counts: torch.Tensor | None = None):
    """Received rows -> halo, local ancestor indices, gather + weight reset + the K moves (one C call)."""
    po, _o = _obs(obs, world.n_rays)
    mk = np.ascontiguousarray(np.asarray(move_keys, np.uint32).reshape(-1, 4, 2))
    if mk.shape[0] > N.MAX_MOVES:
        raise GjxError(f"at most {N.MAX_MOVES} fused moves per call")
    k = {"none": 0, "mh": 1, "hmc": 2}[kind] if mk.shape[0] > 0 else 0
    n_recv = recv.numel() // SHARD_ROW
    srow = _rows(_scores_for(buf, world))
    N.check(N.lib().gjx_shard_resample_moves(
        buf.ctx.handle, C.byref(world), _ptr(buf.cand_idx), _ptr(buf.range), _ptr(recv) if n_recv else C.c_void_p(0),
        n_recv, buf.rank, buf.n, buf.halo, _ptr(buf.anc), _ptr(buf.err), k, mk.ctypes.data_as(C.POINTER(C.c_uint32)),
        mk.shape[0], int(sel_mask), float(step_size), int(n_steps), int(bool(is_initial)), po, buf.ptrs[buf.cur],
        buf.ptrs[1 - buf.cur], _ptr(buf.logw), *srow, _ptr(accept_counts), events[0] if events else None,
        events[1] if events else None, _stream()), "gjx_shard_resample_moves")
    buf.cur = 1 - buf.cur


def _move_key_block(move_keys):
    mk = np.ascontiguousarray(np.asarray(move_keys, np.uint32).reshape(-1, 4, 2))
    if not 1 <= mk.shape[0] <= N.MAX_MOVES:
        raise GjxError(f"between 1 and {N.MAX_MOVES} fused moves per call")
    return mk


def shard_split(buf: ShardStepBuffers, n_moves: int, accept_counts: torch.Tensor | None = None) -> torch.Tensor:
    """After shard_search, still without the host: buf.split = {p0 + 1, p1, go, 0} (own-sourced run [p0, p1) of
    positions, the device's copy of the resampling decision); zeroes the acceptance counters of the step."""
    N.check(N.lib().gjx_shard_split(_ptr(buf.cand_idx), _ptr(buf.range), buf.rank, buf.n, _ptr(buf.gstats),
                                    buf.n_global, _ptr(buf.split), _ptr(accept_counts), int(n_moves), _stream()),
            "gjx_shard_split")
    return buf.split


def shard_moves_main(world: LocWorld, buf: ShardStepBuffers, move_keys, is_initial: bool, obs, sel_mask: int = 7,
                     events=None, accept_counts: torch.Tensor | None = None):
    """The mh moves of every position that needs no row from another rank (of all positions, un-resampled, when the
    device decided against resampling), into the alternate columns; flips buf.cur."""
    po, _o = _obs(obs, world.n_rays)
    mk = _move_key_block(move_keys)
    N.check(N.lib().gjx_shard_moves_main(
        buf.ctx.handle, C.byref(world), _ptr(buf.cand_idx), _ptr(buf.range), buf.rank, buf.n, _ptr(buf.split),
        mk.ctypes.data_as(C.POINTER(C.c_uint32)), mk.shape[0], int(sel_mask), int(bool(is_initial)), po,
        buf.ptrs[buf.cur], buf.ptrs[1 - buf.cur], _ptr(buf.logw), *_rows(_scores_for(buf, world)), _ptr(accept_counts),
        events[0] if events else None, events[1] if events else None, _stream()), "gjx_shard_moves_main")
    buf.cur = 1 - buf.cur


def shard_moves_fixup(world: LocWorld, buf: ShardStepBuffers, recv: torch.Tensor, p0: int, p1: int, move_keys,
                      is_initial: bool, obs, sel_mask: int = 7, accept_counts: torch.Tensor | None = None):
    """After shard_moves_main (which flipped buf.cur) and the all_to_all: received rows -> halo of the PREVIOUS
    column set, then the resampled moves of the positions [0, p0) and [p1, n) into the current one.  Runs on the
    current torch stream, which may differ from the main launch's."""
    po, _o = _obs(obs, world.n_rays)
    mk = _move_key_block(move_keys)
    n_recv = recv.numel() // SHARD_ROW
    N.check(N.lib().gjx_shard_moves_fixup(
        buf.ctx.handle, C.byref(world), _ptr(recv) if n_recv else C.c_void_p(0), n_recv, buf.rank, buf.n, buf.halo,
        int(p0), int(p1), _ptr(buf.anc_fix), _ptr(buf.err), mk.ctypes.data_as(C.POINTER(C.c_uint32)), mk.shape[0],
        int(sel_mask), int(bool(is_initial)), po, buf.ptrs[1 - buf.cur], buf.ptrs[buf.cur], _ptr(buf.logw),
        *_rows(_scores_for(buf, world)), _ptr(accept_counts), _stream()), "gjx_shard_moves_fixup")


def _loc_moves(fn_name, world, move_keys, sel_mask, extra, is_initial, obs, pose_in, means, pose_out, offset,
               accept_counts):
    mk = np.asarray(move_keys, dtype=np.uint32).reshape(-1, 4, 2)
    x_in, y_in, th_in = pose_in
    n = x_in.numel()
    if pose_out is None:
        pose_out = tuple(torch.empty_like(t) for t in pose_in)
    po, _o = _obs(obs, world.n_rays)
    fn = getattr(N.lib(), fn_name)
    src = pose_in
    for s in range(0, mk.shape[0], N.MAX_MOVES):
        chunk = mk[s:s + N.MAX_MOVES]
        pk, _k = _keys_block(chunk, chunk.shape)
        acc = accept_counts[s:s + chunk.shape[0]] if accept_counts is not None else None
        N.check(fn(_ctx_for(x_in).handle, C.byref(world), pk, chunk.shape[0], int(sel_mask), *extra,
                   int(bool(is_initial)), offset, n, po, _ptr(src[0]), _ptr(src[1]), _ptr(src[2]),
                   _ptr(means[0]), _ptr(means[1]), _ptr(means[2]), _ptr(pose_out[0]), _ptr(pose_out[1]),
                   _ptr(pose_out[2]), _ptr(acc), _stream()), fn_name)
        src = pose_out
    return pose_out


def loc_mh(world: LocWorld, move_keys, is_initial: bool, obs, pose_in, means, pose_out=None, sel_mask: int = 7,
           offset: int = 0, accept_counts: torch.Tensor | None = None):
    """n_moves fused MH moves.  move_keys: (n_moves, 4, 2) uint32 = {x', y', theta', u} per move;
    pose_in / means / pose_out are (x, y, th) tuples (pose_out may be pose_in)."""
    return _loc_moves("gjx_loc_mh", world, move_keys, sel_mask, (), is_initial, obs, pose_in, means, pose_out,
                      offset, accept_counts)


def loc_mh_resampled(world: LocWorld, move_keys, is_initial: bool, obs, ancestors: torch.Tensor, src6, dst6=None,
                     sel_mask: int = 7, offset: int = 0, idx_base: int = 0,
                     accept_counts: torch.Tensor | None = None, ray_scores: torch.Tensor | None = None,
                     logw_reset: torch.Tensor | None = None):
    """n_moves fused MH moves on resampled particles: particle i starts from ancestor
    ancestors[i] - idx_base of the six source columns; returns the six new columns.
    ray_scores: optional (n_rays, n_src) stored scores of the source particles (loc_extend_scores)."""
    mk = np.ascontiguousarray(np.asarray(move_keys, dtype=np.uint32).reshape(-1, 4, 2))
    if mk.shape[0] > N.MAX_MOVES:
        raise GjxError(f"at most {N.MAX_MOVES} fused moves per call")
    n = ancestors.numel()
    if dst6 is None:
        dst6 = [torch.empty(n, dtype=torch.float32, device=ancestors.device) for _ in range(6)]
    po, _o = _obs(obs, world.n_rays)
    src_p = (C.c_void_p * 6)(*[_ptr(_f32(t)).value for t in src6])
    dst_p = (C.c_void_p * 6)(*[_ptr(_f32(t)).value for t in dst6])
    ctx = Context.get(ancestors.device.index or 0, n)
    N.check(N.lib().gjx_loc_mh_resampled(ctx.handle, C.byref(world), mk.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         mk.shape[0], int(sel_mask), int(bool(is_initial)), offset, n, po,
                                         _ptr(ancestors), int(idx_base), src_p, dst_p, *_rows(ray_scores),
                                         _ptr(logw_reset),
                                         _ptr(accept_counts), _stream()),
            "gjx_loc_mh_resampled")
    return dst6


def loc_hmc(world: LocWorld, move_keys, step_size: float, n_steps: int, is_initial: bool, obs, pose_in, means,
            pose_out=None, sel_mask: int = 7, offset: int = 0, accept_counts: torch.Tensor | None = None):
    """n_moves fused HMC moves.  move_keys: (n_moves, 4, 2) = {p_theta, p_x, p_y, u} per move."""
    return _loc_moves("gjx_loc_hmc", world, move_keys, sel_mask, (float(step_size), int(n_steps)), is_initial,
                      obs, pose_in, means, pose_out, offset, accept_counts)


def loc_regenerate(world: LocWorld, keys4, is_initial: bool, obs, pose, means, sel_mask: int = 7, offset: int = 0,
                   ray_scores: torch.Tensor | None = None, want_scores: bool = False) -> dict:
    """gjx_loc_regenerate: one regenerate() of every particle.  keys4: (4, 2) = {x', y', theta', u}.
    -> dict(x, y, th proposals, weight, log_u[, scores (n_rays, n)])."""
    x, y, th = (_f32(t).contiguous() for t in pose)
    n = x.numel()
    f = lambda: torch.empty(n, dtype=torch.float32, device=x.device)  # noqa: E731
    out = {"x": f(), "y": f(), "th": f(), "weight": f(), "log_u": f()}
    if want_scores:
        out["scores"] = torch.empty((world.n_rays, n), dtype=torch.float32, device=x.device)
    po, _o = _obs(obs, world.n_rays)
    pk, _k = _keys_block(keys4, (4, 2))
    N.check(N.lib().gjx_loc_regenerate(C.byref(world), pk, int(sel_mask), int(bool(is_initial)), offset, n, po,
                                       _ptr(x), _ptr(y), _ptr(th), _ptr(means[0]), _ptr(means[1]), _ptr(means[2]),
                                       *_rows(ray_scores), _ptr(out["x"]), _ptr(out["y"]), _ptr(out["th"]),
                                       _ptr(out["weight"]), _ptr(out["log_u"]), *_rows(out.get("scores")),
                                       _stream()), "gjx_loc_regenerate")
    return out


def loc_hmc_propose(world: LocWorld, keys4, step_size: float, n_steps: int, is_initial: bool, obs, pose, means,
                    sel_mask: int = 7, offset: int = 0) -> dict:
    """gjx_loc_hmc_propose: the proposal half of one hmc move.  keys4 = {p_theta, p_x, p_y, u}.
    -> dict(x, y, th end points, log_alpha (raw), log_u, logp_cur, logp_new)."""
    x, y, th = (_f32(t).contiguous() for t in pose)
    n = x.numel()
    names = ("x", "y", "th", "log_alpha", "log_u", "logp_cur", "logp_new")
    out = {k_: torch.empty(n, dtype=torch.float32, device=x.device) for k_ in names}
    po, _o = _obs(obs, world.n_rays)
    pk, _k = _keys_block(keys4, (4, 2))
    N.check(N.lib().gjx_loc_hmc_propose(C.byref(world), pk, int(sel_mask), float(step_size), int(n_steps),
                                        int(bool(is_initial)), offset, n, po, _ptr(x), _ptr(y), _ptr(th),
                                        _ptr(means[0]), _ptr(means[1]), _ptr(means[2]),
                                        *[_ptr(out[k_]) for k_ in names], _stream()), "gjx_loc_hmc_propose")
    return out


def loc_logp_grad(world: LocWorld, is_initial: bool, obs, x, y, th, mx, my, mth):
    n = x.numel()
    po, _o = _obs(obs, world.n_rays)
    logp = torch.empty(n, dtype=torch.float32, device=x.device)
    grad = torch.empty((3, n), dtype=torch.float32, device=x.device)
    N.check(N.lib().gjx_loc_logp_grad(C.byref(world), int(bool(is_initial)), n, po, _ptr(x), _ptr(y), _ptr(th),
                                      _ptr(mx), _ptr(my), _ptr(mth), _ptr(logp), _ptr(grad), _stream()),
            "gjx_loc_logp_grad")
    return logp, grad


# --------------------------------------------------------------------------- #
# curvefit polynomial importance sampling
# --------------------------------------------------------------------------- #
def poly_is(keys3, n: int, xs, ys, sigma: float = 0.05, offset: int = 0, device=None):
    xs = np.ascontiguousarray(np.asarray(xs, dtype=np.float32).reshape(-1))
    ys = np.ascontiguousarray(np.asarray(ys, dtype=np.float32).reshape(-1))
    if xs.size != ys.size:
        raise GjxError("xs and ys differ in length")
    dev = _dev(device)
    a, b, c, logw = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4))
    pk, _k = _keys_block(keys3, (3, 2))
    fp = C.POINTER(C.c_float)
    N.check(N.lib().gjx_poly_is(pk, offset, n, xs.ctypes.data_as(fp), ys.ctypes.data_as(fp), xs.size,
                                float(sigma), _ptr(a), _ptr(b), _ptr(c), _ptr(logw), _stream()), "gjx_poly_is")
    return a, b, c, logw


# --------------------------------------------------------------------------- #
# outlier-mixture curve fitting: per-chain enumerative Gibbs + HMC
# --------------------------------------------------------------------------- #
def make_outlier_model(xs, ys, alpha: float = 1.0, beta: float = 10.0, inlier_sigma: float = 0.1,
                       out_lo: float = -2.0, out_hi: float = 2.0) -> OutlierModel:
    xs = np.asarray(xs, np.float32).reshape(-1)
    ys = np.asarray(ys, np.float32).reshape(-1)
    if xs.size != ys.size:
        raise GjxError("xs and ys differ in length")
    if not (1 <= xs.size <= N.MAX_POINTS):
        raise GjxError(f"n_points={xs.size} not in [1, {N.MAX_POINTS}]")
    m = OutlierModel()
    m.n_points = xs.size
    for j in range(xs.size):
        m.xs[j], m.ys[j] = float(xs[j]), float(ys[j])
    m.alpha, m.beta, m.inlier_sigma = float(alpha), float(beta), float(inlier_sigma)
    m.out_lo, m.out_hi = float(out_lo), float(out_hi)
    return m


def outlier_init(model: OutlierModel, base_key, n: int, offset: int = 0, device=None) -> dict:
    """Prior draw per chain -> dict(keys (n, 2) int32 storage, a, b, c, rate, outliers (int64 bit mask))."""
    dev = _dev(device)
    st = {"keys": torch.empty((n, 2), dtype=torch.int32, device=dev),
          "outliers": torch.empty(n, dtype=torch.int64, device=dev)}
    for k_ in ("a", "b", "c", "rate"):
        st[k_] = torch.empty(n, dtype=torch.float32, device=dev)
    N.check(N.lib().gjx_outlier_init(C.byref(model), _ckey(base_key), offset, n, _ptr(st["keys"]), _ptr(st["a"]),
                                     _ptr(st["b"]), _ptr(st["c"]), _ptr(st["rate"]), _ptr(st["outliers"]),
                                     _stream()), "gjx_outlier_init")
    return st


def outlier_gibbs_hmc(model: OutlierModel, state: dict, n_iters: int, step_size: float, n_steps: int,
                      accept_count: torch.Tensor | None = None) -> dict:
    """n_iters fused Gibbs + HMC kernel applications per chain, in place on `state`."""
    n = state["a"].numel()
    N.check(N.lib().gjx_outlier_gibbs_hmc(C.byref(model), int(n_iters), float(step_size), int(n_steps), n,
                                          _ptr(state["keys"]), _ptr(state["a"]), _ptr(state["b"]),
                                          _ptr(state["c"]), _ptr(state["rate"]), _ptr(state["outliers"]),
                                          _ptr(accept_count), _stream()), "gjx_outlier_gibbs_hmc")
    return state


def outlier_gibbs_hmc_diag(model: OutlierModel, state: dict, step_size: float, n_steps: int,
                           accept_count: torch.Tensor | None = None) -> torch.Tensor:
    """ONE Gibbs + HMC application per chain, in place on `state`, with the quantities of its decisions:
    -> (10 + n_points, n) f32 (rows as in gjx_outlier_gibbs_hmc_diag, include/gjx.h)."""
    n = state["a"].numel()
    diag = torch.zeros((10 + int(model.n_points), n), dtype=torch.float32, device=state["a"].device)
    N.check(N.lib().gjx_outlier_gibbs_hmc_diag(C.byref(model), float(step_size), int(n_steps), n,
                                               _ptr(state["keys"]), _ptr(state["a"]), _ptr(state["b"]),
                                               _ptr(state["c"]), _ptr(state["rate"]), _ptr(state["outliers"]),
                                               _ptr(diag), _ptr(accept_count), _stream()), "gjx_outlier_gibbs_hmc_diag")
    return diag


# --------------------------------------------------------------------------- #
# Game-of-Life inversion: batched Gibbs chains on byte boards
# --------------------------------------------------------------------------- #
def _u8(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.uint8:
        raise GjxError(f"expected uint8, got {t.dtype}")
    return t.contiguous()


def _gol_future(future: torch.Tensor, n_chains: int, n: int):
    future = _u8(future)
    if future.shape == (n, n):
        return future, 0
    if future.shape == (n_chains, n, n):
        return future, n * n
    raise GjxError(f"target board must be ({n}, {n}) or ({n_chains}, {n}, {n}), got {tuple(future.shape)}")


def gol_keys(k, n_frames: int, per_frame: int):
    """Host key schedule of one chain -> (init key (2,), pass keys (n_frames, per_frame, 9, 2))."""
    init = _U32x2()
    out = np.zeros((max(n_frames, 0), max(per_frame, 0), 9, 2), np.uint32)
    N.lib().gjx_gol_keys(_ckey(k), n_frames, per_frame, init,
                         out.ctypes.data_as(C.POINTER(C.c_uint32)) if out.size else (C.c_uint32 * 1)())
    return np.array([init[0], init[1]], np.uint32), out


def _dev_keys(keys, dev) -> torch.Tensor:
    a = np.ascontiguousarray(np.asarray(keys, np.uint32))
    return torch.from_numpy(a.view(np.int32)).to(dev)


def gol_init(init_keys, n: int, device=None) -> torch.Tensor:
    """init_keys: (B, 2) uint32 host array -> (B, n, n) uint8 boards."""
    dev = _dev(device)
    keys = _dev_keys(np.asarray(init_keys, np.uint32).reshape(-1, 2), dev)
    b = keys.shape[0]
    boards = torch.empty((b, n, n), dtype=torch.uint8, device=dev)
    N.check(N.lib().gjx_gol_init(_ptr(keys), b, n, _ptr(boards), _stream()), "gjx_gol_init")
    return boards


def gol_gibbs_pass(boards: torch.Tensor, future: torch.Tensor, pass_keys, oy: int, ox: int, p_flip: float):
    boards = _u8(boards)
    b, n, n2 = boards.shape
    if n != n2:
        raise GjxError("Game-of-Life boards must be square")
    fut, stride = _gol_future(future, b, n)
    keys = pass_keys if isinstance(pass_keys, torch.Tensor) else _dev_keys(
        np.asarray(pass_keys, np.uint32).reshape(b, 2), boards.device)
    N.check(N.lib().gjx_gol_gibbs_pass(_ptr(keys), b, n, oy, ox, float(p_flip), _ptr(fut), stride, _ptr(boards),
                                       _stream()), "gjx_gol_gibbs_pass")
    return boards


def gol_sweep(boards: torch.Tensor, future: torch.Tensor, pass_keys9, p_flip: float):
    """One full Gibbs sweep in place; pass_keys9: (9, B, 2) uint32 (host array or int32 device tensor)."""
    boards = _u8(boards)
    b, n, n2 = boards.shape
    if n != n2:
        raise GjxError("Game-of-Life boards must be square")
    fut, stride = _gol_future(future, b, n)
    keys = pass_keys9 if isinstance(pass_keys9, torch.Tensor) else _dev_keys(
        np.asarray(pass_keys9, np.uint32).reshape(9, b, 2), boards.device)
    N.check(N.lib().gjx_gol_sweep(_ptr(keys.contiguous()), b, n, float(p_flip), _ptr(fut), stride, _ptr(boards),
                                  _stream()), "gjx_gol_sweep")
    return boards


def gol_pack(boards: torch.Tensor) -> torch.Tensor:
    """(B, n, n) uint8 boards -> (B, n, ceil(n / 64)) int64 bit-packed boards (bit c of word w = column 64 w + c)."""
    boards = _u8(boards)
    if boards.dim() == 2:
        boards = boards[None]
    b, n, n2 = boards.shape
    if n != n2:
        raise GjxError("Game-of-Life boards must be square")
    bits = torch.empty((b, n, (n + 63) // 64), dtype=torch.int64, device=boards.device)
    N.check(N.lib().gjx_gol_pack(_ptr(boards), b, n, _ptr(bits), _stream()), "gjx_gol_pack")
    return bits


def gol_unpack(bits: torch.Tensor, n: int) -> torch.Tensor:
    b = bits.shape[0]
    if bits.dtype != torch.int64 or tuple(bits.shape[1:]) != (n, (n + 63) // 64):
        raise GjxError("gol_unpack: expected (B, n, ceil(n / 64)) int64 bit-packed boards")
    out = torch.empty((b, n, n), dtype=torch.uint8, device=bits.device)
    N.check(N.lib().gjx_gol_unpack(_ptr(bits), b, n, _ptr(out), _stream()), "gjx_gol_unpack")
    return out


def _gol_future_bits(future_bits: torch.Tensor, b: int, n: int):
    wp = (n + 63) // 64
    if future_bits.dtype != torch.int64 or tuple(future_bits.shape[-2:]) != (n, wp):
        raise GjxError("expected an (n, ceil(n / 64)) int64 bit-packed target")
    if future_bits.dim() == 2 or future_bits.shape[0] == 1:
        return future_bits.contiguous(), 0
    if future_bits.shape[0] != b:
        raise GjxError("per-chain targets must match the number of chains")
    return future_bits.contiguous(), n * wp


def gol_gibbs_pass_packed(bits: torch.Tensor, n: int, future_bits: torch.Tensor, pass_keys, oy: int, ox: int,
                          p_flip: float):
    b = bits.shape[0]
    fut, stride = _gol_future_bits(future_bits, b, n)
    keys = pass_keys if isinstance(pass_keys, torch.Tensor) else _dev_keys(
        np.asarray(pass_keys, np.uint32).reshape(b, 2), bits.device)
    N.check(N.lib().gjx_gol_gibbs_pass_packed(_ptr(keys.contiguous()), b, n, oy, ox, float(p_flip), _ptr(fut), stride,
                                              _ptr(bits), _stream()), "gjx_gol_gibbs_pass_packed")
    return bits


def gol_sweep_packed(bits: torch.Tensor, n: int, future_bits: torch.Tensor, pass_keys9, p_flip: float):
    """One full Gibbs sweep in place on bit-packed boards; pass_keys9: (9, B, 2)."""
    b = bits.shape[0]
    fut, stride = _gol_future_bits(future_bits, b, n)
    keys = pass_keys9 if isinstance(pass_keys9, torch.Tensor) else _dev_keys(
        np.asarray(pass_keys9, np.uint32).reshape(9, b, 2), bits.device)
    N.check(N.lib().gjx_gol_sweep_packed(_ptr(keys.contiguous()), b, n, float(p_flip), _ptr(fut), stride, _ptr(bits),
                                         _stream()), "gjx_gol_sweep_packed")
    return bits


def gol_step(boards: torch.Tensor) -> torch.Tensor:
    boards = _u8(boards)
    b, n, _ = boards.shape
    out = torch.empty_like(boards)
    N.check(N.lib().gjx_gol_step(_ptr(boards), b, n, _ptr(out), _stream()), "gjx_gol_step")
    return out


def gol_summary(boards: torch.Tensor, future: torch.Tensor, p_flip: float):
    """-> (n_incorrect int64 (B,), predictive posterior score float64 (B,))."""
    boards = _u8(boards)
    b, n, _ = boards.shape
    fut, stride = _gol_future(future, b, n)
    bad = torch.empty(b, dtype=torch.int64, device=boards.device)
    sc = torch.empty(b, dtype=torch.float64, device=boards.device)
    N.check(N.lib().gjx_gol_summary(_ptr(boards), _ptr(fut), stride, b, n, float(p_flip), _ptr(bad), _ptr(sc),
                                    _stream()), "gjx_gol_summary")
    return bad, sc


# --------------------------------------------------------------------------- #
# linear-Gaussian state-space family
# --------------------------------------------------------------------------- #
def make_lgssm(initial_mean, initial_cov, A, Q, Cm, R) -> LgssmParams:
    mu = np.asarray(initial_mean, np.float32).reshape(-1)
    d = mu.size
    Cm = np.asarray(Cm, np.float32).reshape(-1, d)
    o = Cm.shape[0]
    if not (1 <= d <= 4 and 1 <= o <= 4):
        raise GjxError(f"linear_gaussian: d_state={d}, d_obs={o} must be in [1, 4]")
    p = LgssmParams()
    p.d_state, p.d_obs = d, o
    for name, arr, shape in (("initial_mean", mu, (d,)), ("initial_cov", initial_cov, (d, d)), ("A", A, (d, d)),
                             ("Q", Q, (d, d)), ("C", Cm, (o, d)), ("R", R, (o, o))):
        a = np.asarray(arr, np.float32).reshape(shape).reshape(-1)
        field = getattr(p, name)
        for i, v in enumerate(a):
            field[i] = float(v)
    return p


def lgssm_extend(model: LgssmParams, k, is_initial: bool, obs, prev, logw_in, offset: int = 0):
    """-> (state (d, n), mean (d, n), logw)."""
    n = logw_in.numel()
    d = model.d_state
    state = torch.empty((d, n), dtype=torch.float32, device=logw_in.device)
    mean = torch.empty_like(state)
    logw = torch.empty_like(logw_in)
    o = np.ascontiguousarray(np.asarray(obs, np.float32).reshape(-1))
    if o.size != model.d_obs:
        raise GjxError(f"observation has {o.size} entries, d_obs={model.d_obs}")
    N.check(N.lib().gjx_lgssm_extend(_ctx_for(logw_in).handle, C.byref(model), _ckey(k), int(bool(is_initial)),
                                     offset, n, o.ctypes.data_as(C.POINTER(C.c_float)), _ptr(prev),
                                     _ptr(logw_in), _ptr(state), _ptr(mean), _ptr(logw), _stream()),
            "gjx_lgssm_extend")
    return state, mean, logw


def lgssm_smc_run(model: LgssmParams, interp_key, obs_rows, n: int, n_moves: int, device=None, want_accepts=False):
    """The whole bootstrap filter (with n_moves mh moves per step when > 0) in one host call; the resampling
    decisions are taken on the device.  -> dict(state, mean, logw, lml (1,), ess (T,), resampled (T,),
    accept_counts (T, n_moves) | None, key = the advanced interpreter key)."""
    dev = _dev(device)
    obs = np.ascontiguousarray(np.asarray(obs_rows, np.float32).reshape(-1, model.d_obs))
    T, ds = obs.shape[0], model.d_state
    f = lambda *shape, dt=torch.float32: torch.empty(shape, dtype=dt, device=dev)  # noqa: E731
    sets = [(f(ds, n), f(ds, n)), (f(ds, n), f(ds, n))]
    logw, stats, lml, trace = f(n), f(5), f(1), f(2 * T)
    go, rng = f(1, dt=torch.int32), f(2, dt=torch.int64)
    acc = torch.zeros((T, n_moves), dtype=torch.int64, device=dev) if (want_accepts and n_moves > 0) else None
    kbuf = _U32x2(*[int(v) for v in np.asarray(interp_key, np.uint32).reshape(2)])
    final = C.c_int32(0)
    ctx = Context.get(dev.index or 0, n)
    N.check(N.lib().gjx_lgssm_smc_run(ctx.handle, C.byref(model), kbuf, T, obs.ctypes.data_as(C.POINTER(C.c_float)), n,
                                      int(n_moves), _ptr(sets[0][0]), _ptr(sets[0][1]), _ptr(sets[1][0]),
                                      _ptr(sets[1][1]), _ptr(logw), _ptr(stats), _ptr(lml), _ptr(trace), _ptr(go),
                                      _ptr(rng), _ptr(acc), C.byref(final), _stream()), "gjx_lgssm_smc_run")
    state, mean = sets[final.value]
    tr = trace.reshape(T, 2)
    return {"state": state, "mean": mean, "logw": logw, "lml": lml, "ess": tr[:, 0], "resampled": tr[:, 1],
            "accept_counts": acc, "key": np.array([kbuf[0], kbuf[1]], np.uint32)}


def lgssm_mh(model: LgssmParams, move_keys, is_initial: bool, obs, state_in, mean, offset: int = 0,
             accept_counts=None):
    """move_keys: (n_moves, 2, 2) = {state', u} per move."""
    mk = np.ascontiguousarray(np.asarray(move_keys, np.uint32).reshape(-1, 2, 2))
    n = state_in.shape[1]
    out = torch.empty_like(state_in)
    o = np.ascontiguousarray(np.asarray(obs, np.float32).reshape(-1))
    N.check(N.lib().gjx_lgssm_mh(_ctx_for(state_in[0]).handle, C.byref(model),
                                 mk.ctypes.data_as(C.POINTER(C.c_uint32)), mk.shape[0], int(bool(is_initial)),
                                 offset, n, o.ctypes.data_as(C.POINTER(C.c_float)), _ptr(state_in), _ptr(mean),
                                 _ptr(out), _ptr(accept_counts), _stream()), "gjx_lgssm_mh")
    return out
