"""genjax.inference surface (reference src/genjax/inference/__init__.py:7-46)."""
from .smc import (  # noqa: F401
    ParticleCollection,
    change,
    effective_sample_size,
    extend,
    extend_csmc,
    init,
    init_csmc,
    resample_vectorized_trace,
    rejuvenate,
    rejuvenation_smc,
    resample,
    systematic_resample,
)
from .mcmc import MCMCResult, chain, compute_ess, compute_rhat, hmc, mala, mh  # noqa: F401
