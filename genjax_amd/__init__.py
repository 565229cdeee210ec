"""genjax_amd -- MI355X-native backend for GenJAX's vectorized-inference hot path.

Public names follow genjax (reference src/genjax/__init__.py:13-132): gen, Fn, Distribution,
Vmap, Tr/Trace, sel/Selection, const/Const, seed, the hot-path distributions, and the
`genjax_amd.inference` module (init, extend, resample, rejuvenate, rejuvenation_smc,
ParticleCollection, mh, hmc)."""
from ._native import GjxError  # noqa: F401
from .core import (  # noqa: F401
    AllSel,
    Cond,
    Const,
    Distribution,
    Fn,
    GFI,
    Selection,
    Tr,
    Trace,
    Vmap,
    batched,
    const,
    distribution,
    gen,
    get_choices,
    modular_vmap,
    get_retval,
    get_score,
    seed,
    sel,
    trace,
)
from .distributions import (  # noqa: F401
    bernoulli,
    beta,
    categorical,
    flip,
    multivariate_normal,
    normal,
    uniform,
)
from . import inference  # noqa: F401
from . import pjax  # noqa: F401
from .backend import fold_in, key, split  # noqa: F401
