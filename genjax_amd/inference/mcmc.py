"""MCMC kernels with the reference's signatures (src/genjax/inference/mcmc.py):
mh(trace, selection) :267-309 and hmc(trace, selection, step_size, n_steps) :434-578.

For traces of a fused model family both dispatch to one fused HIP kernel per call
(regenerate + accept, or momentum + leapfrog with the hand-derived gradient + accept).
For generic `@gen` traces `mh` uses GFI.regenerate on whole columns; `hmc` needs reverse-mode
differentiation of an arbitrary Python model, which only the fused families provide.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import backend as B
from .._native import GjxError
from ..core import Selection, Tr, _next_key, batch_size
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
    shape = (n,) if n is not None else ()
    u = B.sample_uniform(_next_key(), shape, 0.0, 1.0)
    log_alpha = torch.clamp_max(log_weight, 0.0)
    accept = torch.log(u) < log_alpha
    last_accept["value"] = accept
    return _select_trace(accept, new_trace, current_trace, n)


def _select_trace(accept, new, old, n):
    if isinstance(new, Tr):
        return Tr(new._gen_fn, _select_trace(accept, new._args, old._args, n),
                  _select_trace(accept, new._choices, old._choices, n),
                  _select_trace(accept, new._retval, old._retval, n),
                  _select_trace(accept, new._score, old._score, n))
    if isinstance(new, dict):
        return {k: _select_trace(accept, new[k], old[k], n) for k in new}
    if isinstance(new, (tuple, list)):
        return type(new)(_select_trace(accept, a, b, n) for a, b in zip(new, old))
    if isinstance(new, torch.Tensor) and isinstance(old, torch.Tensor):
        a = accept
        while a.dim() < max(new.dim(), old.dim()):
            a = a.unsqueeze(-1)
        return torch.where(a, new, old)
    return new


def hmc(current_trace, selection: Selection, step_size: float, n_steps: int):
    """Single HMC step with leapfrog integration (mcmc.py:434-578)."""
    spec = MoveSpec("hmc", selection, float(step_size), int(n_steps))
    if isinstance(current_trace, ProbeTrace):
        return spec
    if isinstance(current_trace, FusedTrace):
        return _fused_single(current_trace, spec)
    raise GjxError("hmc: gradients are hand-derived per fused model family (SURVEY.md section 7); a generic "
                   "@gen trace has no gradient on this backend")
