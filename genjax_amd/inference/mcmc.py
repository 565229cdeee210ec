"""MCMC kernels with the reference's signatures (src/genjax/inference/mcmc.py):
mh(trace, selection) :267-309 and hmc(trace, selection, step_size, n_steps) :434-578.

For traces of a fused model family both dispatch to one fused HIP kernel per call
(regenerate + accept, or momentum + leapfrog with the hand-derived gradient + accept).
For generic `@gen` traces `mh` uses GFI.regenerate on whole columns, and `hmc` differentiates assess()
through torch.autograd with the library's gradient kernels as the backward of every log-density
(distributions._LogpdfFn); momenta, leapfrog updates, the acceptance test and the selection are library launches.
"""
from __future__ import annotations

import torch

from .. import backend as B
from .._native import GjxError
from ..core import Selection, Tr, _next_key, as_col, batch_size, is_record, record_fields, record_rebuild
from ..families.base import FusedTrace, MoveSpec, ProbeTrace

# last acceptance column / counts, the analogue of state.save(accept=...) (mcmc.py:307,576)
last_accept = {"value": None}


def _fused_single(trace, spec):
    fam = trace.family
    from ..core import _Ambient

    if _Ambient.keys:
        keys = fam.move_keys(spec, _Ambient.keys[-1].key)
        # the move consumed splits of the ambient interpreter: advance it identically
        seq = _Ambient.keys[-1]
        probe = B.KeySeq(seq.key)
        n_splits = _count_splits(fam, spec)
        for _ in range(n_splits):
            probe.next()
        seq.key = probe.key
    else:
        keys = fam.move_keys(spec, _next_key())
    acc = torch.zeros(1, dtype=torch.int64, device=trace.x.device)
    out = fam.fused_moves(trace, spec, keys[None], accept_counts=acc)
    last_accept["value"] = acc
    return out


def _count_splits(fam, spec):
    mask = fam.sel_mask(spec.selection)
    nsel = bin(mask).count("1")
    return nsel + 1 if spec.kind == "mh" else nsel + 2


def mh(current_trace, selection: Selection):
    """Single Metropolis-Hastings step using GFI.regenerate (mcmc.py:267-309)."""
    if isinstance(current_trace, ProbeTrace):
        return MoveSpec("mh", selection)
    if isinstance(current_trace, FusedTrace):
        return _fused_single(current_trace, MoveSpec("mh", selection))
    gf = current_trace.get_gen_fn()
    args = current_trace.get_args()
    new_trace, log_weight, _ = gf.regenerate(current_trace, selection, *args[0], **args[1])
    n = batch_size()
    # log_alpha = min(0, w); accept = log(u) < log_alpha  (mcmc.py:293-296), one launch
    accept8 = B.mh_accept(_next_key(), log_weight, n if n is not None else 1)
    accept = accept8.bool() if n is not None else accept8.bool().reshape(())
    last_accept["value"] = accept
    return _select_trace(accept, new_trace, current_trace, n, accept8)


def _select_trace(accept, new, old, n, accept8=None):
    """jax.lax.select(accept, new_leaf, old_leaf) over every leaf (mcmc.py:298-303)."""
    if isinstance(new, Tr):
        return new._rebuild([_select_trace(accept, a, b, n, accept8) for a, b in zip(new._parts(), old._parts())])
    if isinstance(new, dict):
        return {k: _select_trace(accept, new[k], old[k], n, accept8) for k in new}
    if isinstance(new, (tuple, list)):
        return type(new)(_select_trace(accept, a, b, n, accept8) for a, b in zip(new, old))
    if is_record(new) and is_record(old):
        return record_rebuild(new, [_select_trace(accept, a, b, n, accept8)
                                    for a, b in zip(record_fields(new), record_fields(old))])
    if isinstance(new, torch.Tensor) and isinstance(old, torch.Tensor):
        if new.is_cuda and new.dtype == old.dtype and new.element_size() == 4 and accept8 is not None:
            shp = tuple(torch.broadcast_shapes(tuple(new.shape), tuple(old.shape)))
            if n is not None and len(shp) >= 1 and shp[0] == n:       # per-particle leaves: one flag per row
                return B.select_rows(accept8, new.expand(shp), old.expand(shp))
            if n is None:                                             # one chain: the same flag for every element
                flags = accept8.reshape(1).expand(max(1, new.numel()))
                return B.select_rows(flags.contiguous(), new.expand(shp).reshape(-1), old.expand(shp).reshape(-1)).reshape(shp)
        if n is not None and (new.dim() == 0 or new.shape[0] != n):
            return new              # an argument shared by all particles (unbatched under the reference's vmap)
        # per-particle leaves that are not 4-byte columns (int64 indices, bool flags)
        a = accept
        while a.dim() < max(new.dim(), old.dim()):
            a = a.unsqueeze(-1)
        return torch.where(a, new, old)
    return new


def _leaves(tree, prefix=()):
    """(path, leaf) pairs in jax.tree_util order (dict keys sorted)."""
    if isinstance(tree, dict):
        out = []
        for k in sorted(tree):
            out += _leaves(tree[k], prefix + (k,))
        return out
    return [(prefix, tree)]


def _rebuild(paths, vals):
    out: dict = {}
    for path, v in zip(paths, vals):
        if not path:
            return v
        d = out
        for k in path[:-1]:
            d = d.setdefault(k, {})
        d[path[-1]] = v
    return out


class _SelectedDensity:
    """_create_log_density_wrt_selected (mcmc.py:59-88) on device columns: the selected leaves in tree order
    and log p(merge(unselected, q)) with its gradient (HIP log-density kernels forward, their gradient kernels
    backward through genjax_amd.distributions._LogpdfFn)."""

    def __init__(self, trace, selection):
        self.gf, self.args = trace.get_gen_fn(), trace.get_args()
        self.n = batch_size()
        selected, self.unselected = self.gf.filter(trace.get_choices(), selection)
        self.empty = selected is None
        if not self.empty:
            items = _leaves(selected)
            self.paths = [p for p, _ in items]
            self.q0 = [as_col(v).to(torch.float32) for _, v in items]

    def tree(self, q):
        return _rebuild(self.paths, q)

    def logp(self, q, want_grad):
        """log p and, if wanted, d/dq of its sum over particles (particles are independent)."""
        xs = [v.detach().clone().requires_grad_(want_grad) for v in q]
        with torch.enable_grad() if want_grad else torch.no_grad():
            sel_ = self.tree(xs)
            full = sel_ if self.unselected is None else self.gf.merge(self.unselected, sel_)[0]
            lp, _ = self.gf.assess(full, *self.args[0], **self.args[1])
            if lp.dim() > 1 or lp.numel() != (1 if self.n is None else self.n):
                raise GjxError(f"gradient moves need one log-density per particle, got shape {tuple(lp.shape)}: a "
                               "vector-valued site outside a Vmap scores element-wise (as in the reference); "
                               "put the site under .vmap(...) / repeat(n) so that its score is summed")
            if not want_grad:
                return lp, None
            grads = torch.autograd.grad(lp.sum(), xs, allow_unused=True)
        return lp.detach(), [torch.zeros_like(x) if g is None else g.contiguous() for x, g in zip(xs, grads)]

    def normal_logpdf_total(self, xs, means, sigma):
        """sum over leaves of normal.logpdf(x, mean, sigma), each summed over the leaf's own axes."""
        n, tot = self.n, None
        for x, m in zip(xs, means):
            lp = B.normal_logpdf(x, m, as_col(sigma))
            if n is not None and lp.dim() > 1:
                lp = lp.reshape(n, -1).sum(dim=1)
            elif n is None and lp.dim() > 0:
                lp = lp.sum()
            tot = lp if tot is None else B.sum_diff(tot, lp)
        return tot

    def finish(self, trace, proposed, log_alpha):
        """accept = log(u) < min(0, log_alpha); select leaf by leaf."""
        n = self.n
        accept8 = B.mh_accept(_next_key(), log_alpha, n if n is not None else 1)
        accept = accept8.bool() if n is not None else accept8.bool().reshape(())
        last_accept["value"] = accept
        return _select_trace(accept, proposed, trace, n, accept8)


def _identity_move(trace):
    n = batch_size()
    last_accept["value"] = torch.ones((n,) if n is not None else (), dtype=torch.bool)
    return trace


def _generic_hmc(trace, selection, eps, n_steps):
    """mcmc.py:434-578 for a generic trace.  Key schedule: one site key per selected leaf (momenta, tree order),
    one split for the leapfrog scan, one for the uniform of the acceptance test.
    Pinned against the reference for leaves that are one scalar per particle (the reference draws ONE
    normal.sample(0.0, 1.0) per leaf, mcmc.py:487).  A vector-valued leaf needs a momentum per element: this draws
    them from the leaf's one site key, which the reference does not define -- those draws are covered by a
    distributional test only (tests/test_gpu_surface.py::test_generic_hmc_vector_leaf_is_distributionally_right)."""
    from ..core import _Ambient
    from ..distributions import normal

    d = _SelectedDensity(trace, selection)
    if d.empty:                                            # nothing selected: the move is the identity
        return _identity_move(trace)
    q0 = d.q0
    zeros = [torch.zeros_like(v) for v in q0]
    p = [normal.sample(z, 1.0) for z in zeros]                            # one site key per leaf
    lp0, g = d.logp(q0, True)
    k0 = d.normal_logpdf_total(p, zeros, 1.0)                             # mcmc.py:504-512
    if _Ambient.keys:
        _next_key()                                                        # scan_p over the leapfrog steps: one split
    q = q0
    for _ in range(int(n_steps)):                                          # mcmc.py:395-431
        p = [B.axpy(0.5 * eps, gi, pi) for gi, pi in zip(g, p)]
        q = [B.axpy(eps, pi, qi) for pi, qi in zip(p, q)]
        _, g = d.logp(q, True)
        p = [B.axpy(0.5 * eps, gi, pi) for gi, pi in zip(g, p)]
    proposed, _, _ = d.gf.update(trace, d.tree(q), *d.args[0], **d.args[1])
    lp1, _ = d.logp(q, False)
    k1 = d.normal_logpdf_total(p, zeros, 1.0)        # logpdf(-p) == logpdf(p) bit for bit: the density squares it
    log_alpha = B.sum_diff(lp1, k1, lp0, k0)         # (log p(q') + K(p')) - (log p(q) + K(p))   mcmc.py:556-560
    return d.finish(trace, proposed, log_alpha)


def mala(current_trace, selection: Selection, step_size: float):
    """Single Metropolis-adjusted Langevin step (mcmc.py:312-431): x' = x + (tau^2/2) grad log p(x) + tau eps,
    accepted with the forward / backward proposal densities N(x + drift, tau).  Key schedule: one site key per
    selected leaf (the noise, tree order), then the uniform of the acceptance test."""
    from ..distributions import normal

    if isinstance(current_trace, (ProbeTrace, FusedTrace)):
        raise GjxError("mala: the fused families implement mh and hmc; run mala on a generic @gen trace")
    tau = float(step_size)
    d = _SelectedDensity(current_trace, selection)
    if d.empty:
        return _identity_move(current_trace)
    drift = lambda q, g: [B.axpy(0.5 * tau * tau, gi, qi) for gi, qi in zip(g, q)]     # noqa: E731  mean of the proposal
    _, g0 = d.logp(d.q0, True)
    mean_fwd = drift(d.q0, g0)
    noise = [normal.sample(torch.zeros_like(v), 1.0) for v in d.q0]
    prop = [B.axpy(tau, z, m) for z, m in zip(noise, mean_fwd)]
    forward = d.normal_logpdf_total(prop, mean_fwd, tau)
    proposed, model_weight, _ = d.gf.update(current_trace, d.tree(prop), *d.args[0], **d.args[1])
    _, g1 = d.logp(prop, True)
    backward = d.normal_logpdf_total(d.q0, drift(prop, g1), tau)
    log_alpha = B.sum_diff(model_weight, backward, forward)                # mcmc.py:414
    return d.finish(current_trace, proposed, log_alpha)


def hmc(current_trace, selection: Selection, step_size: float, n_steps: int):
    """Single HMC step with leapfrog integration (mcmc.py:434-578)."""
    spec = MoveSpec("hmc", selection, float(step_size), int(n_steps))
    if isinstance(current_trace, ProbeTrace):
        return spec
    if isinstance(current_trace, FusedTrace):
        return _fused_single(current_trace, spec)
    return _generic_hmc(current_trace, selection, float(step_size), int(n_steps))


# --------------------------------------------------------------------------- #
# chain driver and between-chain diagnostics (mcmc.py:91-247, 581-747)
# --------------------------------------------------------------------------- #
def compute_rhat(samples: torch.Tensor) -> torch.Tensor:
    """Potential scale reduction factor of (n_chains, n_samples) draws (mcmc.py:91-158):
    sqrt(((N - 1) W + B) / N / W); NaN with fewer than two chains."""
    n_chains, n_samples = samples.shape
    if n_chains < 2 or n_samples < 2:
        return torch.tensor(float("nan"), device=samples.device)
    x = samples.double()
    b = n_samples * x.mean(dim=1).var(unbiased=True)
    w = x.var(dim=1, unbiased=True).mean()
    return torch.sqrt(((n_samples - 1) * w + b) / n_samples / w).float()


def compute_ess(samples: torch.Tensor, kind: str = "bulk") -> torch.Tensor:
    """Lag-1 effective sample size (mcmc.py:161-247): M N / (1 + 2 clip(rho_1, 0, 0.99)); the
    "tail" kind uses the per-step spread between the 5 % and 95 % quantiles over chains."""
    n_chains, n_samples = samples.shape
    x = samples.double()
    if kind == "tail":
        q = torch.quantile(x, torch.tensor([0.05, 0.95], dtype=torch.float64, device=x.device), dim=1)
        flat = (q[1] - q[0]).reshape(-1)
    else:
        flat = x.reshape(-1)
    a, b = flat[:-1], flat[1:]
    if a.numel() < 2:
        rho = torch.tensor(float("nan"), dtype=torch.float64, device=x.device)
    else:
        rho = torch.corrcoef(torch.stack([a, b]))[0, 1]
    rho = torch.clamp(rho, 0.0, 0.99)
    total = (n_chains if kind == "bulk" else 1) * n_samples
    return (total / (1 + 2 * rho)).float()


class MCMCResult:
    """traces, accepts, acceptance_rate, n_steps, n_chains, rhat, ess_bulk, ess_tail
    (mcmc.py:250-264).  `traces.get_choices()` leaves are (n_steps, ...) for one chain and
    (n_chains, n_steps, ...) for several."""

    def __init__(self, traces, accepts, acceptance_rate, n_steps, n_chains, rhat=None, ess_bulk=None,
                 ess_tail=None):
        self.traces, self.accepts, self.acceptance_rate = traces, accepts, acceptance_rate
        self.n_steps, self.n_chains = n_steps, n_chains
        self.rhat, self.ess_bulk, self.ess_tail = rhat, ess_bulk, ess_tail


class _ChainTraces:
    """Stack of the traces visited by a chain; choices are stacked on access."""

    def __init__(self, traces, indices, n_chains):
        self._traces, self._indices, self._n_chains = traces, indices, n_chains

    def __len__(self):
        return len(self._indices)

    def __getitem__(self, i):
        return self._traces[self._indices[i]]

    def _stack(self, per_step):
        multi, chains = self._n_chains > 1, self._n_chains

        def stack(items):
            if isinstance(items[0], dict):
                return {k: stack([it[k] for it in items]) for k in items[0]}
            if isinstance(items[0], (tuple, list)):
                return type(items[0])(stack([it[j] for it in items]) for j in range(len(items[0])))
            if is_record(items[0]):
                cols = [record_fields(it) for it in items]
                return record_rebuild(items[0], [stack([c[j] for c in cols]) for j in range(len(cols[0]))])
            if not isinstance(items[0], (torch.Tensor, float, int, bool)):
                return items[-1]
            ts = [torch.as_tensor(v) for v in items]
            if multi:       # a leaf shared by the chains (a constrained site) is written out per chain, as vmap does
                ts = [t if (t.dim() >= 1 and t.shape[0] == chains) else t.expand((chains,) + tuple(t.shape)) for t in ts]
            t = torch.stack(ts)                                            # (steps, [chains], ...)
            return t.transpose(0, 1).contiguous() if multi else t

        return stack(per_step)

    def get_choices(self):
        return self._stack([get_choices_of(self._traces[i]) for i in self._indices])

    def get_retval(self):
        return self._stack([self._traces[i].get_retval() for i in self._indices])

    def get_score(self):
        return self._stack([self._traces[i].get_score() for i in self._indices])


def get_choices_of(trace):
    return trace.get_choices()


def _tree_map(fn, tree):
    if isinstance(tree, dict):
        return {k: _tree_map(fn, v) for k, v in tree.items()}
    return fn(tree)


def _repeat_trace(trace, n):
    """jnp.repeat(leaf[None], n_chains) over every array leaf of the initial trace (mcmc.py:676-679)."""
    if isinstance(trace, Tr):
        return trace._rebuild([_repeat_trace(p, n) for p in trace._parts()])
    if isinstance(trace, dict):
        return {k: _repeat_trace(v, n) for k, v in trace.items()}
    if isinstance(trace, (tuple, list)):
        return type(trace)(_repeat_trace(v, n) for v in trace)
    if is_record(trace):
        return record_rebuild(trace, [_repeat_trace(v, n) for v in record_fields(trace)])
    if isinstance(trace, torch.Tensor):
        return trace.unsqueeze(0).expand((n,) + tuple(trace.shape)).contiguous()
    return trace


def chain(mcmc_kernel):
    """chain(kernel)(initial_trace, n_steps, burn_in=, autocorrelation_resampling=, n_chains=)
    (mcmc.py:581-747): repeated kernel applications with burn-in, thinning and parallel chains.
    Several chains are the particle axis of the kernels (the reference vmaps the single-chain
    program, so all chains draw from the same per-step keys by chain index).  Key schedule under
    seed(): one split for the scan over steps, step i runs on fold_in(sub, i)."""
    from ..core import Const, _Ambient, _push_keys, batched, const

    def run_chain(initial_trace, n_steps, *, burn_in=const(0), autocorrelation_resampling=const(1),
                  n_chains=const(1)):
        val = lambda c: c.value if isinstance(c, Const) else int(c)  # noqa: E731
        steps, burn, thin, chains = val(n_steps), val(burn_in), val(autocorrelation_resampling), val(n_chains)
        fused = isinstance(initial_trace, FusedTrace)
        if fused:
            # a fused trace is a particle batch already: its n particles ARE the chains (one fused launch per step
            # moves all of them); n_chains may only restate that number
            if chains not in (1, initial_trace.n):
                raise GjxError(f"chain: a fused trace of {initial_trace.n} particles runs {initial_trace.n} chains; "
                               f"n_chains={chains} cannot be honoured (build the trace with that many particles)")
            chains = initial_trace.n
            trace = initial_trace
        else:
            trace = _repeat_trace(initial_trace, chains) if chains > 1 else initial_trace
        seeded = bool(_Ambient.keys)
        sub = _next_key() if seeded else None                      # scan_p: one split
        traces, accepts = [], []
        for i in range(steps):
            scope = _push_keys(B.fold_in(sub, i)) if seeded else None
            if scope:
                scope.__enter__()
            try:
                if chains > 1 or fused:
                    with batched(chains):
                        trace = mcmc_kernel(trace)
                else:
                    trace = mcmc_kernel(trace)
            finally:
                if scope:
                    scope.__exit__(None, None, None)
            traces.append(trace)
            acc = last_accept["value"]
            if fused:      # the fused kernels count acceptances (one int64 per move): the per-step rate over chains
                acc = (acc.double() / chains).float().reshape(()).expand(chains) if chains > 1 else acc.float().reshape(())
            accepts.append(torch.as_tensor(acc).float() if acc is not None else torch.zeros(()))
        idx = list(range(burn, steps, thin))
        acc_all = torch.stack(accepts) if accepts else torch.zeros(0)
        final_acc = acc_all[idx]                                   # (steps', [chains])
        result_traces = _ChainTraces(traces, idx, chains)
        if chains == 1:
            return MCMCResult(result_traces, final_acc, final_acc.mean(), const(len(idx)), const(1))
        final_acc = final_acc.transpose(0, 1)                      # (chains, steps')
        rates = final_acc.mean(dim=1)
        choices = result_traces.get_choices()

        def diag(fn):
            return _tree_map(lambda s_: fn(s_) if isinstance(s_, torch.Tensor) and s_.dim() == 2
                             else torch.tensor(float("nan")), choices)

        return MCMCResult(result_traces, final_acc, rates.mean(), const(len(idx)), const(chains),
                          rhat=diag(compute_rhat), ess_bulk=diag(lambda s_: compute_ess(s_, "bulk")),
                          ess_tail=diag(lambda s_: compute_ess(s_, "tail")))

    return run_chain
