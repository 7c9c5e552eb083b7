"""Drop-in for the reference's activation.py (trunc_exp)."""
from focnerf_amd.activation import trunc_exp  # noqa: F401
