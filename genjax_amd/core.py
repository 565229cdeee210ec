"""Generative-function interface (GFI) of the reference, re-hosted on device columns.

Mirrors the call surface of genjax.core (reference src/genjax/core.py): `gen`, `Fn`,
`Distribution`, `Vmap`, `Tr`, `sel`/`Selection`, `const`/`Const` with the same method names,
argument order, weight algebra and error behaviour:

  simulate(*args) -> Tr                                  core.py:1584-1591, 2136-2146
  generate(x | None, *args) -> (Tr, weight)              core.py:1593-1604, 2148-2163
  assess(x, *args) -> (logp, retval)                     core.py:1606-1613, 2165-2176
  update(tr, x_, *args) -> (Tr, weight, discard)         core.py:1615-1636, 2178-2198
  regenerate(tr, sel, *args) -> (Tr, weight, discard)    core.py:1638-1657, 2200-2219
  Tr.get_score() is -log p                               core.py:673-677

There is no tracing compiler here.  The reference obtains its particle axis by jax.vmap-ing a
scalar program; this package runs the user's `@gen` body ONCE on whole columns: every value
is a float32 device tensor whose leading axis is the particle axis (or a scalar that
broadcasts).  Sampling sites draw a whole column from one sub-key with the HIP Threefry
kernel and log-densities are evaluated by the HIP log-density kernels, so the per-site key
schedule and the draw layout are those of `seed` + `modular_vmap` (pjax.py:860-908,
1333-1338).  torch is used for buffers and for the user's own deterministic glue
(e.g. `a + b * x`), never for sampling or density evaluation.

Case-study models with fused kernels (localization, curvefit, linear-Gaussian) derive from
`FusedFamily` in genjax_amd.families and bypass this generic path.
"""
from __future__ import annotations

import functools
from dataclasses import dataclass
from typing import Any, Callable

import numpy as np
import torch

from . import backend as B
from ._native import GjxError


# --------------------------------------------------------------------------- #
# Const (reference core.py:199-442): static, non-traced values
# --------------------------------------------------------------------------- #
class Const:
    """Wrapper marking a value as static (reference `Const[...]`, core.py:199)."""

    __slots__ = ("value",)

    def __init__(self, value):
        self.value = value

    def __repr__(self):
        return f"Const({self.value!r})"

    def __eq__(self, other):
        return isinstance(other, Const) and self.value == other.value

    def __hash__(self):
        return hash(("Const", self.value if isinstance(self.value, (int, float, str, tuple, bool)) else id(self.value)))

    def _binop(self, other, op):
        o = other.value if isinstance(other, Const) else other
        return Const(op(self.value, o))

    __add__ = lambda s, o: s._binop(o, lambda a, b: a + b)
    __sub__ = lambda s, o: s._binop(o, lambda a, b: a - b)
    __mul__ = lambda s, o: s._binop(o, lambda a, b: a * b)
    __floordiv__ = lambda s, o: s._binop(o, lambda a, b: a // b)


def const(v) -> Const:
    return Const(v)


# --------------------------------------------------------------------------- #
# Selections (reference core.py:781-1018)
# --------------------------------------------------------------------------- #
class AllSel:
    """Matches every address (reference core.py:895-905)."""


class NoneSel:
    """Matches no address (reference core.py:882-892)."""


class Selection:
    """Address filter with the reference's algebra: `|` union, `^` intersection, `~` complement,
    `match(addr) -> (bool, sub-selection)`, `() in sel` for leaf sites."""

    def __init__(self, kind, payload=None):
        if isinstance(kind, AllSel):                 # Selection(AllSel()), the reference's spelling (core.py:895)
            kind = "all"
        elif isinstance(kind, NoneSel):
            kind = "none"
        self.kind, self.payload = kind, payload

    def match(self, addr):
        k, p = self.kind, self.payload
        if k == "all":
            return True, self
        if k == "none":
            return False, self
        if k == "str":
            hit = addr == p
            return hit, Selection("all") if hit else Selection("none")
        if k == "path":
            if not p:
                return False, Selection("none")
            hit = addr == p[0]
            if not hit:
                return False, Selection("none")
            return True, (Selection("all") if len(p) == 1 else Selection("path", p[1:]))
        if k == "dict":
            if addr in p:
                sub = p[addr]
                return True, sub if isinstance(sub, Selection) else sel(sub)
            return False, Selection("none")
        if k == "not":
            hit, rest = p.match(addr)
            return (not hit), Selection("not", rest)
        if k == "and":
            h1, r1 = p[0].match(addr)
            h2, r2 = p[1].match(addr)
            return (h1 and h2), Selection("and", (r1, r2))
        if k == "or":
            h1, r1 = p[0].match(addr)
            h2, r2 = p[1].match(addr)
            return (h1 or h2), Selection("or", (r1, r2))
        raise GjxError(f"bad selection kind {k}")

    def __call__(self, addr):
        return self.match(addr)

    def __contains__(self, addr):
        return self.match(addr)[0]

    def __or__(self, other):
        return Selection("or", (self, other))

    def __xor__(self, other):
        return Selection("and", (self, other))

    def __invert__(self):
        return Selection("not", self)

    def __repr__(self):
        return f"Selection({self.kind}, {self.payload!r})"


def sel(*v) -> Selection:
    """sel("x"), sel(("outer","inner")), sel(()), sel({"a": sel("b")}), sel()/sel(None)."""
    assert len(v) <= 1
    if not v or v[0] is None:
        return Selection("none")
    a = v[0]
    if isinstance(a, tuple) and len(a) == 0:
        return Selection("all")
    if isinstance(a, dict):
        return Selection("dict", a)
    if isinstance(a, tuple) and all(isinstance(s, str) for s in a):
        return Selection("path", tuple(a))
    assert isinstance(a, str)
    return Selection("str", a)


# --------------------------------------------------------------------------- #
# ambient interpreter state: key schedule and particle axis
# --------------------------------------------------------------------------- #
class _Ambient:
    keys: list = []          # stack of KeySeq (innermost last)
    batch: list = []         # stack of particle counts
    inner: list = []         # stack of Vmap axis sizes: sites under a Vmap draw along a trailing axis of that size
    counter = 0              # reference's global counter for un-seeded calls (pjax.py:723-811)


def _next_key() -> np.ndarray:
    if _Ambient.keys:
        return _Ambient.keys[-1].next()
    _Ambient.counter += 1     # KeylessWrapper: jrand.key(global_counter.count)
    return B.key(_Ambient.counter)


class _push_keys:
    def __init__(self, k):
        self.seq = k if isinstance(k, B.KeySeq) else B.KeySeq(k)

    def __enter__(self):
        _Ambient.keys.append(self.seq)
        return self.seq

    def __exit__(self, *exc):
        _Ambient.keys.pop()


class batched:
    """Context: run GFI methods over `n` particles at once (stands in for modular_vmap with
    axis_size=n, pjax.py:1649-1655)."""

    def __init__(self, n: int):
        self.n = int(n)

    def __enter__(self):
        _Ambient.batch.append(self.n)
        return self

    def __exit__(self, *exc):
        _Ambient.batch.pop()


def batch_size():
    return _Ambient.batch[-1] if _Ambient.batch else None


def modular_vmap(f: Callable, in_axes=0, axis_size=None, axis_name=None, spmd_axis_name=None) -> Callable:
    """modular_vmap(f, in_axes, axis_size)(*args) (reference pjax.py:1649-1655): run `f` over a leading axis.
    Columns carry that axis already (mapped arguments are (n, ...) tensors, unmapped ones are shared), so the
    transform is the particle-axis context: every sampling site inside draws n values from its one key."""

    def first_len(a, ax):
        if ax is None:
            return None
        if isinstance(a, dict):
            a = list(a.values())
        if isinstance(a, (tuple, list)):
            axs = ax if isinstance(ax, (tuple, list)) else (ax,) * len(a)
            for u, v in zip(a, axs):
                n = first_len(u, v)
                if n is not None:
                    return n
            return None
        return int(a.shape[0]) if isinstance(a, torch.Tensor) and a.dim() >= 1 else None

    @functools.wraps(f)
    def wrapped(*args, **kwargs):
        n = axis_size
        if n is None:
            axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
            n = first_len(args, tuple(axes))
        if n is None:
            raise ValueError("modular_vmap: no mapped argument with a leading axis and no axis_size")
        with batched(int(n)):
            return f(*args, **kwargs)

    return wrapped


def seed(f: Callable) -> Callable:
    """seed(f)(key, *args): run `f` with an explicit PRNG key (reference pjax.py:1414-1455).
    Every sampling site then takes `key, sub = split(key)` in program order."""

    @functools.wraps(f)
    def wrapped(key, *args, **kwargs):
        with _push_keys(key):
            return f(*args, **kwargs)

    return wrapped


def _dev():
    return B._dev()


def as_col(v) -> torch.Tensor:
    """Lift python / numpy values to float32 device tensors."""
    if isinstance(v, torch.Tensor):
        return v if v.is_cuda else v.to(_dev())
    return torch.as_tensor(np.asarray(v, dtype=np.float32), device=_dev())


def _zero():
    return torch.zeros((), dtype=torch.float32, device=_dev())


# --------------------------------------------------------------------------- #
# traces
# --------------------------------------------------------------------------- #
def get_choices(x):
    """Strip Tr wrappers from a (possibly nested) choice structure (reference core.py:get_choices)."""
    if isinstance(x, Tr):
        return get_choices(x.get_choices())
    if isinstance(x, dict):
        return {k: get_choices(v) for k, v in x.items()}
    return x


def is_record(x) -> bool:
    """A user's dataclass instance (the reference's Pytree.dataclass values, e.g. a Pose of columns): the tree
    walks of the inference library descend into its fields."""
    import dataclasses

    return (dataclasses.is_dataclass(x) and not isinstance(x, type) and not isinstance(x, Tr)
            and not getattr(x, "__gjx_static__", False))      # a static record (a world) is a leaf


def record_fields(x) -> list:
    import dataclasses

    return [getattr(x, f.name) for f in dataclasses.fields(x)]


def record_rebuild(x, values):
    import dataclasses

    return dataclasses.replace(x, **{f.name: v for f, v in zip(dataclasses.fields(x), values)})


def get_score(x):
    return x.get_score()


def get_retval(x):
    return x.get_retval()


@dataclass
class Tr:
    """Trace (reference core.py:635-677): gen_fn, args, choices, retval, score = -log p."""

    _gen_fn: Any
    _args: Any
    _choices: Any
    _retval: Any
    _score: Any

    def get_gen_fn(self):
        return self._gen_fn

    def get_args(self):
        return self._args

    def get_choices(self):
        return get_choices(self._choices)

    def get_retval(self):
        return self._retval

    _vm_axis: Any = None        # set by Vmap: size of the mapped axis this trace's leaves carry LAST

    def get_score(self):
        # the reference sums shaped scores (core.py:673-677); here the particle axis (first) is kept
        s = self._score
        if _Ambient.inner:
            # read while a Vmap evaluates its callee: the mapped axis stays (a Cond selects element-wise on it,
            # the enclosing Vmap sums it); a nested Vmap's own trace gives up its own axis only
            own = self._vm_axis is not None and isinstance(s, torch.Tensor) and s.dim() >= 1 and s.shape[-1] == self._vm_axis
            return s.sum(dim=-1) if own else s
        return reduce_inner_axes(s, self._vm_axis)

    def get_fixed_choices(self):
        return self._choices

    # tree structure for the leaf maps of the inference library (gather, select, repeat)
    def _parts(self):
        return [self._args, self._choices, self._retval, self._score]

    def _rebuild(self, parts):
        return Tr(self._gen_fn, *parts, _vm_axis=self._vm_axis)


Trace = Tr


def reduce_inner_axes(s, vm_axis=None):
    """jnp.sum of a score / weight over everything but the particle axis.  A tensor whose LAST axis is a Vmap's
    mapped axis and that has no other axis does not depend on the particle: it sums to one shared scalar."""
    if not isinstance(s, torch.Tensor) or s.dim() == 0:
        return s
    if vm_axis is not None and s.dim() == 1 and s.shape[0] == vm_axis:
        return s.sum()
    if s.dim() > 1:
        return s.reshape(s.shape[0], -1).sum(dim=1)
    return s


# --------------------------------------------------------------------------- #
# GFI base
# --------------------------------------------------------------------------- #
class _Thunk:
    def __init__(self, gen_fn, args, kwargs):
        self.gen_fn, self.args, self.kwargs = gen_fn, args, kwargs

    def __matmul__(self, addr: str):
        return trace(addr, self.gen_fn, self.args, self.kwargs)


class GFI:
    """Generative function interface (reference core.py:1025-1361)."""

    def __call__(self, *args, **kwargs):
        return _Thunk(self, args, kwargs)

    def vmap(self, in_axes=0, axis_size=None, axis_name=None, spmd_axis_name=None):
        return Vmap(self, const(in_axes), const(axis_size), const(axis_name), const(spmd_axis_name))

    def repeat(self, n: int):
        """n independent copies along a new axis (reference core.py:1319-1320)."""
        return self.vmap(in_axes=None, axis_size=n)

    def cond(self, callee_):
        return Cond(self, callee_)

    def log_density(self, x, *args, **kwargs):
        lp = self.assess(x, *args, **kwargs)[0]
        return reduce_inner_axes(lp) if batch_size() is not None else (lp.sum() if isinstance(lp, torch.Tensor) and lp.dim() else lp)

    # the seven GFI methods are provided by subclasses
    def simulate(self, *args, **kwargs):
        raise NotImplementedError

    def generate(self, x, *args, **kwargs):
        raise NotImplementedError

    def assess(self, x, *args, **kwargs):
        raise NotImplementedError

    def update(self, tr, x_, *args, **kwargs):
        raise NotImplementedError

    def regenerate(self, tr, s, *args, **kwargs):
        raise NotImplementedError

    def merge(self, x, x_, check=None):
        raise NotImplementedError

    def filter(self, x, selection):
        raise NotImplementedError


# --------------------------------------------------------------------------- #
# Distribution
# --------------------------------------------------------------------------- #
class Distribution(GFI):
    """A distribution = (sampler, logpdf) pair implementing the GFI (reference core.py:1537-1695).

    `sampler(key, shape, *args)` draws a full column from ONE key; `logpdf(x, *args)` is
    element-wise over the column.  Both run in libgjx_hip.so."""

    def __init__(self, sampler, logpdf, name=None, event_ndims: int = 0):
        self._sampler, self._logpdf, self.name, self.event_ndims = sampler, logpdf, name, event_ndims

    # -- shapes: broadcast of the parameters, with the particle axis in front
    def _shape(self, args):
        shapes = []
        for a in args:
            if isinstance(a, torch.Tensor):
                shapes.append(tuple(a.shape[: a.dim() - self._param_event_ndims(a)]))
        shp = tuple(torch.broadcast_shapes(*shapes)) if shapes else ()
        if _Ambient.inner and _Ambient.inner[-1] is not None:
            # under a Vmap every instance of the mapped axis draws its own value, also at a site whose parameters
            # do not vary along that axis (repeat(n), in_axes None)
            A = _Ambient.inner[-1]
            if len(shp) == 0 or shp[-1] not in (1, A):
                shp = shp + (A,)
            elif shp[-1] == 1:
                shp = shp[:-1] + (A,)
        n = batch_size()
        if n is not None and (len(shp) == 0 or shp[0] != n):
            shp = (n,) + shp
        return shp

    def _param_event_ndims(self, a):
        return 0

    def sample(self, *args, **kwargs):
        args = tuple(as_col(a) if not isinstance(a, (int, np.integer)) or True else a for a in args)
        return self._sampler(_next_key(), self._shape(args), *args, **kwargs)

    def logpdf(self, x, *args, **kwargs):
        return self._logpdf(as_col(x) if not isinstance(x, torch.Tensor) else x, *args, **kwargs)

    def simulate(self, *args, **kwargs):
        x = self.sample(*args, **kwargs)
        lp = self.logpdf(x, *args, **kwargs)
        return Tr(self, (args, kwargs), x, x, -lp)

    def generate(self, x, *args, **kwargs):
        if x is None:
            return self.simulate(*args, **kwargs), _zero()
        x = as_col(x)             # choices are device columns (a python scalar broadcasts)
        lp, _ = self.assess(x, *args, **kwargs)
        return Tr(self, (args, kwargs), x, x, -lp), lp

    def assess(self, x, *args, **kwargs):
        return self.logpdf(x, *args, **kwargs), x

    def update(self, tr, x_, *args, **kwargs):
        x_ = get_choices(tr) if x_ is None else as_col(x_)
        lp = self.logpdf(x_, *args, **kwargs)
        return Tr(self, (args, kwargs), x_, x_, -lp), lp + tr._score, tr.get_retval()   # element-wise: a Vmap sums

    def regenerate(self, tr, s, *args, **kwargs):
        if () in s:
            return self.simulate(*args, **kwargs), _zero(), get_choices(tr)
        x_ = get_choices(tr)
        lp = self.logpdf(x_, *args, **kwargs)
        return Tr(self, (args, kwargs), x_, x_, -lp), lp + tr._score, None

    def merge(self, x, x_, check=None):
        if check is None:
            raise Exception("Can't merge: the underlying sample space `X` for the type `Distribution` "
                            "doesn't support merging without a check parameter.")
        return torch.where(check, x, x_), None

    def filter(self, x, selection):
        hit, _ = selection.match(())
        return (x, None) if hit else (None, x)


class _UserDistribution(Distribution):
    """distribution(sampler, logpdf): both callables take the distribution's arguments only, as the reference's
    do (core.py:1697-1720); a sampler that draws does so through this package's distributions (ambient key)."""

    def sample(self, *args, **kwargs):
        return as_col(self._sampler(*args, **kwargs))

    def logpdf(self, x, *args, **kwargs):
        return as_col(self._logpdf(as_col(x), *args, **kwargs))


def distribution(sampler, logpdf, /, name=None) -> Distribution:
    return _UserDistribution(sampler, logpdf, name)


# --------------------------------------------------------------------------- #
# handlers + Fn (@gen)     reference core.py:1910-2311
# --------------------------------------------------------------------------- #
_handlers: list = []


def _collision(addr, seen, fn):
    if addr in seen:
        raise ValueError(f"Address collision detected: '{addr}' is used multiple times at the same level "
                         f"in generative function {getattr(fn, '__name__', fn)!r}. "
                         "Each address in a generative function must be unique.")


class _Handler:
    def __init__(self, mode, fn, trace=None, choices=None, selection=None):
        self.mode, self.fn, self.trace, self.choices, self.selection = mode, fn, trace, choices, selection
        self.score, self.weight = _zero(), _zero()
        self.trace_map: dict = {}
        self.discard: dict = {}
        self.visited: set = set()

    def __call__(self, addr, gen_fn, args, kwargs):
        m = self.mode
        if m == "assess":
            _collision(addr, self.visited, self.fn)
            self.visited.add(addr)
            lp, r = gen_fn.assess(get_choices(self.choices[addr]), *args, **kwargs)
            self.weight = self.weight + lp
            return r
        _collision(addr, self.trace_map, self.fn)
        if m == "simulate":
            tr = gen_fn.simulate(*args, **kwargs)
            w = None
        elif m == "generate":
            x = get_choices(self.choices[addr]) if addr in self.choices else None
            tr, w = gen_fn.generate(x, *args, **kwargs)
        elif m == "update":
            sub = self.trace._choices[addr]
            x = self.choices[addr] if addr in self.choices else self.trace.get_choices()[addr]
            tr, w, d = gen_fn.update(sub, get_choices(x), *args, **kwargs)
            self.discard[addr] = d
        elif m == "regenerate":
            sub = self.trace._choices[addr]
            _, subsel = self.selection.match(addr)
            tr, w, d = gen_fn.regenerate(sub, subsel, *args, **kwargs)
            self.discard[addr] = d
        else:
            raise GjxError(f"unknown handler mode {m}")
        self.score = self.score + tr.get_score()
        if w is not None:
            self.weight = self.weight + w
        self.trace_map[addr] = tr
        return tr.get_retval()


def trace(addr: str, gen_fn: GFI, args, kwargs=None):
    if not _handlers:
        raise GjxError("`gen_fn(...) @ addr` used outside a @gen function")
    return _handlers[-1](addr, gen_fn, args, kwargs or {})


class Fn(GFI):
    """A generative function made from a Python function with `@gen` (reference core.py:2097)."""

    def __init__(self, source: Callable):
        self.source = Const(source)
        self.__name__ = getattr(source, "__name__", "gen_fn")

    def _run(self, handler, args, kwargs):
        _handlers.append(handler)
        try:
            r = self.source.value(*args, **kwargs)
        finally:
            _handlers.pop()
        return r

    def simulate(self, *args, **kwargs):
        h = _Handler("simulate", self)
        r = self._run(h, args, kwargs)
        return Tr(self, (args, kwargs), h.trace_map, r, h.score)

    def generate(self, x, *args, **kwargs):
        if x is None:
            return self.simulate(*args, **kwargs), _zero()
        h = _Handler("generate", self, choices=x)
        r = self._run(h, args, kwargs)
        return Tr(self, (args, kwargs), h.trace_map, r, h.score), h.weight

    def assess(self, x, *args, **kwargs):
        h = _Handler("assess", self, choices=x)
        r = self._run(h, args, kwargs)
        return h.weight, r

    def update(self, tr, x_, *args, **kwargs):
        h = _Handler("update", self, trace=tr, choices={} if x_ is None else x_)
        r = self._run(h, args, kwargs)
        return Tr(self, (args, kwargs), h.trace_map, r, h.score), h.weight, h.discard

    def regenerate(self, tr, s, *args, **kwargs):
        h = _Handler("regenerate", self, trace=tr, selection=s)
        r = self._run(h, args, kwargs)
        return Tr(self, (args, kwargs), h.trace_map, r, h.score), h.weight, h.discard

    def merge(self, x, x_, check=None):
        out, dropped = {}, {}
        for k in set(x) | set(x_):
            if k in x and k in x_:
                a, b = x[k], x_[k]
                if isinstance(a, dict) and isinstance(b, dict):
                    out[k], d = self.merge(a, b, check)
                    if d is not None:
                        dropped[k] = d
                elif check is not None:
                    out[k] = torch.where(check, a, b)
                else:  # x_ wins, x is discarded (reference core.py:2252-2255)
                    out[k], dropped[k] = b, a
            else:
                out[k] = x[k] if k in x else x_[k]
        return out, (dropped or None)

    def filter(self, x, selection):
        if not x:
            return None, None
        picked, rest = {}, {}
        for addr, v in x.items():
            hit, sub = selection.match(addr)
            if hit and isinstance(v, dict) and sub is not None:
                p, r = self.filter(v, sub)
                if p is not None:
                    picked[addr] = p
                if r is not None:
                    rest[addr] = r
            elif hit:
                picked[addr] = v
            else:
                rest[addr] = v
        return (picked or None), (rest or None)


def gen(fn: Callable) -> Fn:
    """Decorator: Python function -> generative function (reference core.py:2314)."""
    return Fn(fn)


# --------------------------------------------------------------------------- #
# Vmap combinator (reference core.py:1364-1529)
# --------------------------------------------------------------------------- #
class Vmap(GFI):
    """Vectorise a generative function over an inner axis.

    Column semantics: mapped arguments (in_axes entry 0) carry the inner axis LAST
    (shape (..., A) or (N, A)); unmapped column arguments get a trailing unit axis so that
    they broadcast.  The inner function runs once on the widened columns and the weights are
    summed over the inner axis, as the reference does with jnp.sum (core.py:1433,1452,1472,1492).
    """

    def __init__(self, gen_fn, in_axes=Const(0), axis_size=Const(None), axis_name=Const(None),
                 spmd_axis_name=Const(None)):
        self.gen_fn = gen_fn
        self.in_axes = in_axes if isinstance(in_axes, Const) else Const(in_axes)
        self.axis_size = axis_size if isinstance(axis_size, Const) else Const(axis_size)
        self.axis_name, self.spmd_axis_name = axis_name, spmd_axis_name

    def _axes(self, nargs):
        ia = self.in_axes.value
        if ia is None:
            return (None,) * nargs
        if isinstance(ia, int):
            return (ia,) * nargs
        return tuple(ia)

    def _widen(self, args):
        axes = self._axes(len(args))
        out = []
        for a, ax in zip(args, axes):
            if ax is None and isinstance(a, torch.Tensor) and a.dim() >= 1 and batch_size() is not None \
                    and a.shape[0] == batch_size():
                out.append(a.unsqueeze(-1))
            elif ax is None and callable(a) and hasattr(a, "unsqueeze_params"):
                out.append(a.unsqueeze_params())
            else:
                out.append(as_col(a) if (ax is not None and not isinstance(a, torch.Tensor)) else a)
        return tuple(out)

    def _axis(self, args):
        """Size of the mapped axis: axis_size when given, else the last dimension of the first mapped argument."""
        if self.axis_size.value is not None:
            return int(self.axis_size.value)
        for a, ax in zip(args, self._axes(len(args))):
            if ax is not None and isinstance(a, torch.Tensor) and a.dim() >= 1:
                return int(a.shape[-1])
        return None

    def _sum(self, w, A):
        if _Ambient.inner:          # nested under another Vmap: give up this Vmap's axis only
            return w.sum(dim=-1) if isinstance(w, torch.Tensor) and w.dim() >= 1 and w.shape[-1] == A else w
        if isinstance(w, torch.Tensor) and w.dim() >= 1 and batch_size() is None:
            return w.sum()
        return reduce_inner_axes(w, A)

    @staticmethod
    def _own(tr, A):
        if isinstance(tr, Tr):
            tr._vm_axis = A
        return tr

    class _inside:
        def __init__(self, A):
            self.A = A

        def __enter__(self):
            _Ambient.inner.append(self.A)

        def __exit__(self, *exc):
            _Ambient.inner.pop()

    def simulate(self, *args, **kwargs):
        A = self._axis(args)
        with self._inside(A):
            tr = self.gen_fn.simulate(*self._widen(args), **kwargs)
        return self._own(tr, A)

    def generate(self, x, *args, **kwargs):
        A = self._axis(args)
        with self._inside(A):
            tr, w = self.gen_fn.generate(x, *self._widen(args), **kwargs)
        return self._own(tr, A), self._sum(w, A)

    def assess(self, x, *args, **kwargs):
        A = self._axis(args)
        with self._inside(A):
            lp, r = self.gen_fn.assess(x, *self._widen(args), **kwargs)
        return self._sum(lp, A), r

    def update(self, tr, x_, *args, **kwargs):
        A = self._axis(args)
        with self._inside(A):
            new, w, d = self.gen_fn.update(tr, x_, *self._widen(args), **kwargs)
        return self._own(new, A), self._sum(w, A), d

    def regenerate(self, tr, s, *args, **kwargs):
        A = self._axis(args)
        with self._inside(A):
            new, w, d = self.gen_fn.regenerate(tr, s, *self._widen(args), **kwargs)
        return self._own(new, A), self._sum(w, A), d

    def merge(self, x, x_, check=None):
        return self.gen_fn.merge(x, x_, check)

    def filter(self, x, selection):
        return self.gen_fn.filter(x, selection)


# --------------------------------------------------------------------------- #
# Cond combinator (reference core.py:2638-2785)
# --------------------------------------------------------------------------- #
def select_where(check, a, b):
    """jnp.where(check, a, b) on device columns.  4-byte leaves go through gjx_select_rows_b32: per row when
    `check` is one flag per particle, per element otherwise."""
    if isinstance(a, dict) and isinstance(b, dict):
        # addresses only one branch writes are kept as they are (the reference's merge with a check, core.py:2221-2262)
        return {k: (select_where(check, a[k], b[k]) if (k in a and k in b) else (a[k] if k in a else b[k]))
                for k in list(a) + [k for k in b if k not in a]}
    if isinstance(a, (tuple, list)):
        return type(a)(select_where(check, u, v) for u, v in zip(a, b))
    if is_record(a) and is_record(b):
        return record_rebuild(a, [select_where(check, u, v) for u, v in zip(record_fields(a), record_fields(b))])
    if a is None or b is None:
        return a if b is None else b
    if not isinstance(a, (torch.Tensor, float, int, bool, np.ndarray, np.generic)):
        return a                      # static leaves (a world, a Const): the same object on both sides
    a, b = as_col(a), as_col(b)
    check = check if isinstance(check, torch.Tensor) else torch.as_tensor(check, device=a.device)
    shp = tuple(torch.broadcast_shapes(tuple(check.shape), tuple(a.shape), tuple(b.shape)))
    if a.dtype == b.dtype and a.element_size() == 4 and a.is_cuda and not (a.requires_grad or b.requires_grad):
        ae, be = a.expand(shp).contiguous(), b.expand(shp).contiguous()
        c8 = (check != 0).to(torch.uint8)
        if len(shp) >= 1 and tuple(c8.shape) == shp[:1]:
            return B.select_rows(c8.contiguous(), ae, be)
        flat = B.select_rows(c8.expand(shp).contiguous().reshape(-1), ae.reshape(-1), be.reshape(-1))
        return flat.reshape(shp)
    return torch.where(check.bool(), a, b)      # differentiated by autograd (generic hmc), or a host scalar


class CondTr(Tr):
    """Trace of a Cond (reference core.py:2638-2668): the check and both branch traces; every accessor is
    where(check, first, second)."""

    def __init__(self, gen_fn, check, trs):
        self._gen_fn, self.check, self.trs = gen_fn, check, list(trs)

    _args = property(lambda self: ((self.check,) + tuple(self.trs[0].get_args()[0]), self.trs[0].get_args()[1]))
    _choices = property(lambda self: select_where(self.check, self.trs[0].get_choices(), self.trs[1].get_choices()))
    _retval = property(lambda self: select_where(self.check, self.trs[0].get_retval(), self.trs[1].get_retval()))
    _score = property(lambda self: select_where(self.check, self.trs[0].get_score(), self.trs[1].get_score()))

    def _parts(self):
        return [self.check, self.trs[0], self.trs[1]]

    def _rebuild(self, parts):
        return CondTr(self._gen_fn, parts[0], parts[1:])


class Cond(GFI):
    """Cond(callee, callee_)(check, *args): the branch `callee` where check is true, `callee_` elsewhere.
    BOTH branches are evaluated and the results selected (reference core.py:2638-2785); the key schedule is the
    two branches' sites in order (no split for the branch itself: no lax.cond is involved)."""

    def __init__(self, callee, callee_):
        self.callee, self.callee_ = callee, callee_

    def merge(self, x, x_, check=None):
        return self.callee.merge(x, x_, check)

    def filter(self, x, selection):
        return self.callee.filter(x, selection)

    def simulate(self, check, *args, **kwargs):
        return CondTr(self, check, [self.callee.simulate(*args, **kwargs), self.callee_.simulate(*args, **kwargs)])

    def assess(self, x, check, *args, **kwargs):
        lp, r = self.callee.assess(x, *args, **kwargs)
        lp_, r_ = self.callee_.assess(x, *args, **kwargs)
        return select_where(check, lp, lp_), select_where(check, r, r_)

    def generate(self, x, check, *args, **kwargs):
        if x is None:
            return self.simulate(check, *args, **kwargs), _zero()
        tr, w = self.callee.generate(x, *args, **kwargs)
        tr_, w_ = self.callee_.generate(x, *args, **kwargs)
        return CondTr(self, check, [tr, tr_]), select_where(check, w, w_)

    def update(self, tr, x, check, *args, **kwargs):
        new, w, d = self.callee.update(tr.trs[0], x, *args, **kwargs)
        new_, w_, d_ = self.callee_.update(tr.trs[1], x, *args, **kwargs)
        merged = d_ if d is None else (d if d_ is None else self.callee.merge(d, d_)[0])
        return CondTr(self, check, [new, new_]), select_where(check, w, w_), merged

    def regenerate(self, tr, s, check, *args, **kwargs):
        new, w, d = self.callee.regenerate(tr.trs[0], s, *args, **kwargs)
        new_, w_, d_ = self.callee_.regenerate(tr.trs[1], s, *args, **kwargs)
        merged = d_ if d is None else (d if d_ is None else self.callee.merge(d, d_)[0])
        return CondTr(self, check, [new, new_]), select_where(check, w, w_), merged
