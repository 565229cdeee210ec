"""Model families with fused HIP kernels (the registry of SURVEY.md section 7 step 2)."""
from .base import FusedFamily, FusedTrace, MoveSpec  # noqa: F401
