"""genjax.pjax names used on the inference path (reference src/genjax/pjax.py): seed (:1414-1455) and
modular_vmap (:1649-1655), plus the key helpers the tests build keys with.  The reference implements them as
JAX interpreters; here they are the ambient key schedule and the particle-axis context of genjax_amd.core."""
from .backend import fold_in, key, split  # noqa: F401
from .core import modular_vmap, seed  # noqa: F401
