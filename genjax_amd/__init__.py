"""genjax_amd -- MI355X-native backend for GenJAX's vectorized-inference hot path."""
from ._native import GjxError  # noqa: F401
