"""Sequential Monte Carlo on device columns: the reference's genjax.inference.smc call surface
(src/genjax/inference/smc.py) with every per-particle stage in libgjx_hip.so.

  effective_sample_size   smc.py:24-76      two-pass max / sum-exp kernels
  systematic_resample     smc.py:79-136     association-exact cumsum + bisection kernels
  ParticleCollection      smc.py:178-245
  init / extend           smc.py:272-371, 449-530   fused generate kernels (or the generic column path)
  rejuvenate              smc.py:533-600    fused mh / hmc kernels
  resample                smc.py:603-652    lse, ancestors, one gather launch for all columns
  change                  smc.py:374-446    generate of the new target under the renamed choices
  init_csmc / extend_csmc smc.py:788-945    retained particle 0 (generic @gen surface)
  rejuvenation_smc        smc.py:655-785    driver with the Seed key schedule (pjax.py:1318-1401)
"""
from __future__ import annotations

import contextlib
import gc
import math
from typing import Any, Callable

import numpy as np
import torch

from .. import backend as B
from .._native import GjxError
from ..core import (Const, Tr, _Ambient, _next_key, _push_keys, batched, const, is_record, record_fields,
                    record_rebuild, select_where)
from ..families.base import FusedFamily, FusedTrace, MoveSpec, ProbeTrace


# --------------------------------------------------------------------------- #
# weights
# --------------------------------------------------------------------------- #
def effective_sample_size(log_weights: torch.Tensor) -> torch.Tensor:
    """ESS = 1 / sum(softmax(logw)^2)  (smc.py:24-76) -> 0-dim device tensor."""
    return B.logsumexp_ess(log_weights)[2]


def _host_uniform01(key) -> float:
    """uniform.sample(0.0, 1.0) for a scalar site: the f32 value jax.random.uniform would draw,
    computed on the host from the Threefry block (integer + exact float ops only)."""
    import ctypes as C

    from .. import _native as N

    out = (C.c_uint32 * 2)()
    N.lib().gjx_threefry2x32(B._ckey(key), (C.c_uint32 * 2)(0, 0), out)
    bits = np.uint32(out[0] ^ out[1])
    f = (np.uint32((bits >> np.uint32(9)) | np.uint32(0x3F800000))).view(np.float32) - np.float32(1.0)
    return float(f)


def systematic_resample(log_weights: torch.Tensor, n_samples: int) -> torch.Tensor:
    """Ancestor indices by systematic resampling (smc.py:79-136); n_samples must equal len(log_weights)
    as at the reference's only call site (smc.py:165)."""
    n = log_weights.numel()
    if n_samples != n:
        raise GjxError("systematic_resample: n_samples must equal the number of particles")
    stats = B.logsumexp_ess(log_weights)
    w = B.normalized_weights(log_weights, stats[1:2])
    u = _host_uniform01(_next_key())
    return B.resample_systematic(w, u)


# --------------------------------------------------------------------------- #
# particle collection
# --------------------------------------------------------------------------- #
class ParticleCollection:
    """traces, log_weights, diagnostic_weights, n_samples: Const[int], log_marginal_estimate
    (smc.py:178-245)."""

    def __init__(self, traces, log_weights, n_samples, log_marginal_estimate=None, diagnostic_weights=None):
        self.traces = traces
        self.log_weights = log_weights
        self.n_samples = n_samples if isinstance(n_samples, Const) else const(int(n_samples))
        if log_marginal_estimate is None:
            log_marginal_estimate = torch.zeros((), dtype=torch.float32, device=log_weights.device)
        self.log_marginal_estimate = log_marginal_estimate
        self._diag = diagnostic_weights
        self._stats = None

    def _norm_stats(self):
        if self._stats is None:
            self._stats = B.logsumexp_ess(self.log_weights)
        return self._stats

    @property
    def diagnostic_weights(self):
        # log normalised weights (smc.py:259-261); materialised on first access
        if self._diag is None:
            self._diag = B.sub_scalar(self.log_weights, self._norm_stats()[1:2])
        elif callable(self._diag):                     # deferred by resample()
            self._diag = self._diag()
        return self._diag

    def effective_sample_size(self):
        return self._norm_stats()[2]

    def log_marginal_likelihood(self):
        """log_marginal_estimate + logsumexp(logw) - log N  (smc.py:193-204)."""
        n = self.n_samples.value
        log_n = np.float32(math.log(n)) if n > 0 else np.float32(-math.inf)        # the empty collection
        return self.log_marginal_estimate + (self._norm_stats()[1] - log_n)

    def estimate(self, fn: Callable[[Any], Any]):
        """Self-normalised weighted mean of fn(choices) (smc.py:206-245)."""
        values = fn(self.traces.get_choices())
        return B.weighted_sum(self.log_weights, self._norm_stats()[1:2], values)      # one launch, f64 partials


# --------------------------------------------------------------------------- #
# generic trace helpers
# --------------------------------------------------------------------------- #
def _map_leaves(tree, n, fn):
    if isinstance(tree, Tr):
        return tree._rebuild([_map_leaves(p, n, fn) for p in tree._parts()])
    if isinstance(tree, dict):
        return {k: _map_leaves(v, n, fn) for k, v in tree.items()}
    if isinstance(tree, (tuple, list)):
        return type(tree)(_map_leaves(v, n, fn) for v in tree)
    if is_record(tree):
        return record_rebuild(tree, [_map_leaves(v, n, fn) for v in record_fields(tree)])
    if isinstance(tree, torch.Tensor) and tree.dim() >= 1 and tree.shape[0] == n:
        return fn(tree)
    return tree


def _gather_trace(trace, ancestors, n):
    if isinstance(trace, FusedTrace):
        return trace.gather(ancestors)
    idx64 = ancestors.long()

    def take(t):
        if t.dim() == 1 and t.dtype == torch.float32:
            return B.gather_columns(ancestors, [t])[0]
        return t.index_select(0, idx64)

    return _map_leaves(trace, n, take)


def resample_vectorized_trace(trace, log_weights: torch.Tensor, n_samples: int, method: str = "categorical"):
    """Draw ancestor indices from the weights and index every particle-axis leaf of the trace with them
    (smc.py:139-175)."""
    if method == "categorical":
        ancestors = B.resample_multinomial(_next_key(), log_weights, int(n_samples))
    elif method == "systematic":
        ancestors = systematic_resample(log_weights, int(n_samples))
    else:
        raise ValueError(f"Unknown resampling method: {method}")
    return _gather_trace(trace, ancestors, log_weights.numel())


# --------------------------------------------------------------------------- #
# init / extend / rejuvenate / resample
# --------------------------------------------------------------------------- #
def init(target_gf, target_args: tuple, n_samples: Const, constraints, proposal_gf=None) -> ParticleCollection:
    """Importance-sampling initialisation (smc.py:272-371)."""
    n = n_samples.value if isinstance(n_samples, Const) else int(n_samples)
    if isinstance(target_gf, FusedFamily) and proposal_gf is None:
        traces, logw = target_gf.fused_init(tuple(target_args), n, constraints)
    else:
        with batched(n):
            if proposal_gf is None:
                traces, logw = target_gf.generate(constraints, *target_args)
            else:
                # proposal signature (constraints, *target_args)  (smc.py:332-350)
                ptr = proposal_gf.simulate(constraints, *target_args)
                merged, _ = target_gf.merge(ptr.get_choices(), constraints)
                traces, tw = target_gf.generate(merged, *target_args)
                logw = B.sum_diff(tw, ptr.get_score())
        logw = logw.expand(n).contiguous() if logw.dim() == 0 else logw
    return ParticleCollection(traces, logw, const(n))


def extend(particles: ParticleCollection, extended_target_gf, extended_target_args, constraints,
           extension_proposal=None) -> ParticleCollection:
    """Extension move (smc.py:449-530): logw += generate weight (+ proposal score)."""
    n = particles.n_samples.value
    if isinstance(extended_target_gf, FusedFamily) and extension_proposal is None:
        traces, logw = extended_target_gf.fused_extend(particles.traces, particles.log_weights,
                                                       extended_target_args, constraints)
    elif isinstance(extended_target_gf, FusedFamily) and hasattr(extension_proposal, "fused_extend"):
        traces, logw = extension_proposal.fused_extend(extended_target_gf, particles.traces, particles.log_weights,
                                                       extended_target_args, constraints)
    else:
        args = extended_target_args if isinstance(extended_target_args, tuple) else (extended_target_args,)
        with batched(n):
            if extension_proposal is None:
                traces, w = extended_target_gf.generate(constraints, *args)
                logw = B.sum_diff(particles.log_weights, w)
            else:
                # proposal gets (obs, prev_particle_choices, *model_args)  (smc.py:494-509)
                old_choices = particles.traces.get_choices()
                etr = extension_proposal.simulate(constraints, old_choices, *args)
                merged, _ = extended_target_gf.merge(constraints, etr.get_choices())
                traces, w = extended_target_gf.generate(merged, *args)
                logw = B.sum_diff(B.sum_diff(particles.log_weights, w), etr.get_score())
    return ParticleCollection(traces, logw, particles.n_samples, particles.log_marginal_estimate)


def change(particles: ParticleCollection, new_target_gf, new_target_args: tuple, choice_fn: Callable) -> ParticleCollection:
    """Change-of-target move (smc.py:374-446): the particles' choices, renamed by choice_fn (a bijection on
    addresses that leaves the values alone), constrain a generate of the new target; logw += its weight."""
    if isinstance(new_target_gf, FusedFamily) or isinstance(particles.traces, FusedTrace):
        raise GjxError("change: fused families keep their columns in family layout; use the generic @gen surface")
    n = particles.n_samples.value
    with batched(n):
        mapped = choice_fn(particles.traces.get_choices())
        traces, w = new_target_gf.generate(mapped, *new_target_args)
        logw = B.sum_diff(particles.log_weights, w)
    return ParticleCollection(traces, logw, particles.n_samples, particles.log_marginal_estimate)


def _retained_first(n, device):
    flag = torch.zeros(n, dtype=torch.uint8, device=device)
    flag[:1] = 1
    return flag


def init_csmc(target_gf, target_args: tuple, n_samples: Const, constraints, retained_choices,
              proposal_gf=None) -> ParticleCollection:
    """Conditional-SMC initialisation (smc.py:788-839): an ordinary init whose particle 0 carries the
    log-density of the retained choices as its weight (the reference does not overwrite its choices)."""
    n = n_samples.value if isinstance(n_samples, Const) else int(n_samples)
    if n < 1:
        raise ValueError("n_samples must be at least 1 for conditional SMC")
    particles = init(target_gf, target_args, n_samples, constraints, proposal_gf)
    retained_density, _ = target_gf.assess(retained_choices, *target_args)
    lw = particles.log_weights
    logw = select_where(_retained_first(n, lw.device), B._f32(torch.as_tensor(retained_density, device=lw.device)).reshape(-1)[:1].expand(n), lw)
    return ParticleCollection(particles.traces, logw, particles.n_samples, particles.log_marginal_estimate,
                              diagnostic_weights=particles.diagnostic_weights)


def extend_csmc(particles: ParticleCollection, extended_target_gf, extended_target_args, constraints,
                retained_choices, extension_proposal=None) -> ParticleCollection:
    """Extension for conditional SMC (smc.py:842-945): particle 0 takes retained_choices with weight
    old + log-density (trace score = -log-density), every other particle the ordinary extension."""
    n = particles.n_samples.value
    regular = extend(particles, extended_target_gf, extended_target_args, constraints, extension_proposal)
    args = extended_target_args if isinstance(extended_target_args, tuple) else (extended_target_args,)
    first = _retained_first(n, particles.log_weights.device)
    with batched(n):
        density, _ = extended_target_gf.assess(retained_choices, *args)
        # same values as the reference's hand-built Tr (choices = retained_choices, score = -log-density) in the
        # target's own trace layout, so that the two branches select leaf by leaf
        kept, _ = extended_target_gf.generate(retained_choices, *args)
        density = B._f32(torch.as_tensor(density, device=first.device))
        kept_w = B.sum_diff(particles.log_weights, density.expand(n).contiguous() if density.dim() == 0 else density)
    traces = _select_traces(first, kept, regular.traces, n)
    logw = select_where(first, kept_w, regular.log_weights)
    return ParticleCollection(traces, logw, particles.n_samples, particles.log_marginal_estimate)


def _select_traces(flag, a, b, n):
    """where(flag, a, b) over two traces of the same generative function, leaf by leaf."""
    if isinstance(b, Tr):
        return b._rebuild([_select_traces(flag, u, v, n) for u, v in zip(a._parts(), b._parts())])
    if isinstance(b, dict):
        return {k: _select_traces(flag, a[k], v, n) for k, v in b.items()}
    if isinstance(b, (tuple, list)):
        return type(b)(_select_traces(flag, u, v, n) for u, v in zip(a, b))
    if is_record(b):
        return record_rebuild(b, [_select_traces(flag, u, v, n) for u, v in zip(record_fields(a), record_fields(b))])
    if isinstance(b, torch.Tensor) or isinstance(a, torch.Tensor):
        dev = flag.device
        b = b if isinstance(b, torch.Tensor) else torch.as_tensor(b, device=dev)
        a = torch.as_tensor(a, dtype=b.dtype, device=dev)
        per_particle = lambda t: t.dim() >= 1 and t.shape[0] == n      # noqa: E731
        if not per_particle(a) and not per_particle(b):
            if a.shape == b.shape and torch.equal(a, b):
                return b                                                  # shared by every particle in both
            a, b = a.expand((n,) + tuple(a.shape)), b.expand((n,) + tuple(b.shape))
        shape = tuple(b.shape) if per_particle(b) else tuple(a.shape)
        return select_where(flag, a.expand(shape).contiguous(), b.expand(shape).contiguous())
    return b


def rejuvenate(particles: ParticleCollection, mcmc_kernel: Callable) -> ParticleCollection:
    """Apply an MCMC kernel to every particle; weights unchanged (smc.py:533-600)."""
    with batched(particles.n_samples.value):
        new_traces = mcmc_kernel(particles.traces)
    return ParticleCollection(new_traces, particles.log_weights, particles.n_samples,
                              particles.log_marginal_estimate, particles._diag if particles._diag is not None
                              else None)


def resample(particles: ParticleCollection, method: str = "categorical") -> ParticleCollection:
    """Resample (smc.py:603-652): lml += lse - log N, gather every column, weights := 0."""
    n = particles.n_samples.value
    stats = particles._norm_stats()
    old_logw = particles.log_weights
    log_normalized = lambda: B.sub_scalar(old_logw, stats[1:2])    # noqa: E731  pre-resample normalised weights (smc.py:628)
    new_lml = particles.log_marginal_estimate + (stats[1] - np.float32(math.log(n)))
    if method == "categorical":
        ancestors = B.resample_multinomial(_next_key(), particles.log_weights, n)
    elif method == "systematic":
        u = _host_uniform01(_next_key())
        ancestors = B.resample_systematic_logw(particles.log_weights, stats[1:2], u)   # smc.py:127-134
    else:
        raise ValueError(f"Unknown resampling method: {method}")
    traces = _gather_trace(particles.traces, ancestors, n)
    return ParticleCollection(traces, torch.zeros_like(particles.log_weights), particles.n_samples, new_lml,
                              diagnostic_weights=log_normalized)


# --------------------------------------------------------------------------- #
# rejuvenation_smc
# --------------------------------------------------------------------------- #
def _tree_index(tree, i):
    if isinstance(tree, dict):
        return {k: _tree_index(v, i) for k, v in tree.items()}
    return tree[i]


def _tree_len(tree):
    if isinstance(tree, dict):
        return _tree_len(next(iter(tree.values())))
    return len(tree)


class _Scope:
    """A fresh Seed interpreter on `key` when the run is seeded; otherwise a no-op
    (un-seeded runs use the global counter, pjax.py:804-811)."""

    def __init__(self, key):
        self.ctx = _push_keys(key) if key is not None else None

    def __enter__(self):
        if self.ctx:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx:
            self.ctx.__exit__(*exc)


def _seeded():
    return bool(_Ambient.keys)


DEVICE_DECISIONS = True     # False: always the stage-by-stage driver with a host decision per step (tests)


def _probe_kernel(kernel, model):
    """Learn which move `kernel(trace)` performs by calling it once on a probe trace."""
    if not isinstance(model, FusedFamily):
        return None
    try:
        res = kernel(ProbeTrace(model))
    except Exception:
        return None
    return res if isinstance(res, MoveSpec) else None


@contextlib.contextmanager
def _no_gc_pauses():
    """The time loop enqueues a few launches per step and reads one scalar back; a full pass of
    CPython's cyclic GC in the middle of it (35-40 ms with torch loaded) leaves the GPU idle for many
    steps.  Collections are postponed to the end of the loop; the previous setting is restored."""
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


def rejuvenation_smc(model, transition_proposal=None, mcmc_kernel=None, observations=None,
                     initial_model_args=None, n_particles: Const = const(1000),
                     return_all_particles: Const = const(False), n_rejuvenation_moves: Const = const(1),
                     resample_method: Const = const("categorical")):
    """SMC with rejuvenation (smc.py:655-785): init, then per observation extend -> ESS ->
    resample iff ESS < N // 2 -> K rejuvenation moves.  The model's return value feeds the next
    step's arguments.

    `resample_method` is the one addition to the reference signature: the reference hard-codes
    resample(p) = "categorical" (smc.py:712-717), an O(N^2) Gumbel-max that cannot run at
    N >= 2^17; pass const("systematic") for large N (finding 4).

    Key schedule when run under seed(...) (pjax.py:1318-1401): each cond and scan takes one
    split; scan iteration i runs a fresh interpreter on fold_in(sub, i).
    """
    n = n_particles.value
    K = n_rejuvenation_moves.value
    method = resample_method.value if isinstance(resample_method, Const) else resample_method
    kernel = mcmc_kernel.value if isinstance(mcmc_kernel, Const) else mcmc_kernel
    spec = _probe_kernel(kernel, model) if kernel is not None else None
    T = _tree_len(observations)
    # families whose step never needs the host (the lax.cond decided on the device) run the whole filter in one call
    whole = getattr(model, "fused_smc_run", None) if DEVICE_DECISIONS else None
    if (whole is not None and _seeded() and transition_proposal is None and method == "systematic"
            and not return_all_particles.value and (kernel is None or (spec is not None and K > 0))):
        done = whole(observations, initial_model_args, n, K if kernel is not None else 0, spec)
        if done is not None:
            return done
    history = []
    diagnostics = {"ess": [], "resampled": [], "accept_counts": []}

    def maybe_resample(p):
        kc = _next_key() if _seeded() else None          # cond_p: one split, taken or not
        ess = float(p.effective_sample_size().item())     # host decision = lax.cond
        diagnostics["ess"].append(ess)
        take = ess < (n // 2)
        diagnostics["resampled"].append(bool(take))
        if take:
            with _Scope(kc):
                p = resample(p, method)
        return p

    def rejuvenation(p):
        if kernel is None:
            return p
        sr = _next_key() if _seeded() else None           # scan_p: one split
        if spec is not None and isinstance(p.traces, FusedTrace):
            if K > 0 and sr is not None and hasattr(model, "move_keys_block"):
                keys = model.move_keys_block(spec, sr, K)      # the scan's key schedule in one host call
            else:
                keys = np.stack([model.move_keys(spec, B.fold_in(sr, m) if sr is not None else _next_key())
                                 for m in range(K)]) if K > 0 else None
            if K > 0:
                acc = torch.zeros(K, dtype=torch.int64, device=p.log_weights.device)
                tr = model.fused_moves(p.traces, spec, keys, accept_counts=acc)
                diagnostics["accept_counts"].append(acc)
                p = ParticleCollection(tr, p.log_weights, p.n_samples, p.log_marginal_estimate, p._diag)
            return p
        for m in range(K):
            with _Scope(B.fold_in(sr, m) if sr is not None else None):
                p = rejuvenate(p, kernel)
        return p

    particles = init(model, initial_model_args, n_particles, _tree_index(observations, 0))
    particles = maybe_resample(particles)
    particles = rejuvenation(particles)
    if return_all_particles.value:
        history.append(particles)

    st = _next_key() if _seeded() else None               # scan over observations[1:]
    with _no_gc_pauses():
        for t in range(1, T):
            with _Scope(B.fold_in(st, t - 1) if st is not None else None):
                retvals = particles.traces.get_retval()
                particles = extend(particles, model, retvals, _tree_index(observations, t),
                                   extension_proposal=transition_proposal)
                particles = maybe_resample(particles)
                particles = rejuvenation(particles)
            if return_all_particles.value:
                history.append(particles)

    particles.diagnostics = diagnostics
    if return_all_particles.value:
        return StackedParticles(history, diagnostics)
    return particles


class StackedParticles:
    """return_all_particles=True result: a ParticleCollection with a leading time axis
    (smc.py:775-783).  Columns are stacked on access."""

    def __init__(self, steps, diagnostics=None):
        self.steps = steps
        self.n_samples = steps[0].n_samples
        self.diagnostics = diagnostics

    @property
    def log_weights(self):
        return torch.stack([p.log_weights for p in self.steps])

    @property
    def diagnostic_weights(self):
        return torch.stack([p.diagnostic_weights for p in self.steps])

    @property
    def log_marginal_estimate(self):
        return torch.stack([p.log_marginal_estimate for p in self.steps])

    @property
    def traces(self):
        return _StackedTraces([p.traces for p in self.steps])

    def __getitem__(self, t):
        return self.steps[t]

    def __len__(self):
        return len(self.steps)


class _StackedTraces:
    def __init__(self, traces):
        self._traces = traces

    def get_choices(self):
        per_step = [t.get_choices() for t in self._traces]

        def stack(items):
            if isinstance(items[0], dict):
                return {k: stack([it[k] for it in items]) for k in items[0]}
            # a site constrained to one shared value is held once per step; the reference's vmap writes it out
            # per particle, so the stacked leaf has the particle axis either way
            shape = tuple(torch.broadcast_shapes(*[tuple(torch.as_tensor(it).shape) for it in items]))
            return torch.stack([torch.as_tensor(it).expand(shape) for it in items])

        return stack(per_step)
