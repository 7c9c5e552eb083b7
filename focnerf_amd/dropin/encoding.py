"""The `encoding` module the reference's networks import but do not ship."""
from focnerf_amd.encoding import get_encoder  # noqa: F401
