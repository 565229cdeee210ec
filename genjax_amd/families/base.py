"""Base classes of the fused model families.

A `FusedFamily` is a generative function (it has the GFI method names of the reference,
core.py:1025-1361) whose particle-batched generate / mh / hmc are single fused HIP kernels
instead of a traced Python body.  `genjax_amd.inference` dispatches on it.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any

from ..core import GFI, Selection
from .._native import GjxError


class FusedFamily(GFI):
    """Marker + protocol for model families backed by fused kernels."""

    name = "fused"

    # particle-batched entry points used by genjax_amd.inference.smc
    def fused_init(self, args: tuple, n: int, constraints, offset: int = 0):
        """-> (FusedTrace, log_weights) for init() with the default proposal (smc.py:296-317)."""
        raise NotImplementedError

    def fused_extend(self, trace, log_weights, args, constraints, offset: int = 0):
        """-> (FusedTrace, log_weights) for extend() with the default proposal (smc.py:486-491)."""
        raise NotImplementedError

    def fused_moves(self, trace, spec: "MoveSpec", move_keys, offset: int = 0):
        """Apply len(move_keys) MCMC moves in one launch -> (FusedTrace, accept_counts)."""
        raise NotImplementedError

    # The reference's model as a plain @gen function over columns, when the family has one: the GFI methods the
    # fused kernels do not cover (a constrained latent, a user-written proposal, update / regenerate on a generic
    # trace) run through it, so that every call pattern of the reference's surface is served.
    generic: Any = None

    def _twin(self, what: str):
        if self.generic is None:
            self.unsupported(what)
        return self.generic

    def simulate(self, *args, **kwargs):
        return self._twin("simulate").simulate(*args, **kwargs)

    def generate(self, x, *args, **kwargs):
        return self._twin("generate").generate(x, *args, **kwargs)

    def assess(self, x, *args, **kwargs):
        return self._twin("assess").assess(x, *args, **kwargs)

    def update(self, tr, x_, *args, **kwargs):
        return self._twin("update").update(tr, x_, *args, **kwargs)

    def regenerate(self, tr, s, *args, **kwargs):
        return self._twin("regenerate").regenerate(tr, s, *args, **kwargs)

    def merge(self, x, x_, check=None):
        return self._twin("merge").merge(x, x_, check)

    def filter(self, x, selection):
        return self._twin("filter").filter(x, selection)

    def unsupported(self, what: str):
        raise GjxError(f"{self.name}: {what} is not provided by the fused family; write the model with "
                       "@gen (generic column path) for that call pattern")


class FusedTrace:
    """Particle-batched trace over SoA columns; the accessor names are the reference's
    (core.py:656-677)."""

    family: FusedFamily

    def get_gen_fn(self):
        return self.family

    def gather(self, ancestors, idx_base: int = 0) -> "FusedTrace":
        raise NotImplementedError

    def columns(self) -> dict:
        raise NotImplementedError


@dataclass
class MoveSpec:
    """What an `mcmc_kernel(trace)` closure asks for, discovered by probing it once."""

    kind: str            # "mh" | "hmc"
    selection: Selection
    step_size: float = 0.0
    n_steps: int = 0


class ProbeTrace(FusedTrace):
    """Stand-in trace handed to a user's mcmc_kernel to learn which move it performs."""

    def __init__(self, family):
        self.family = family
        self.is_probe = True
