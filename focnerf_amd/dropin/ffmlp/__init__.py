"""Drop-in for the reference's `ffmlp` package (`from ffmlp import FFMLP`)."""
from focnerf_amd.ffmlp import FFMLP, ffmlp_forward  # noqa: F401
