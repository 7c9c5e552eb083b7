"""Drop-in for the reference's `gridencoder` package (`from gridencoder import GridEncoder`)."""
from focnerf_amd.gridencoder import GridEncoder, grid_encode  # noqa: F401
