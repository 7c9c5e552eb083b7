"""focnerf_amd — MI355X (gfx950) implementation of FOCNeRF's volume-rendering hot path:
raymarching, gridencoder, freqencoder, ffmlp (+ the multi-object combine), behind the
reference's Python operator API. The compute lives in libfocnerf_hip.so (C ABI in
include/focnerf.h); importing this package fails loudly if that library is missing."""
from . import _lib  # noqa: F401  (raises ImportError when libfocnerf_hip.so is absent)

__all__ = ["raymarching", "gridencoder", "freqencoder", "ffmlp", "activation", "encoding", "shencoder",
           "renderer", "network", "combine"]
