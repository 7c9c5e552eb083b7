"""Drop-in for the reference's `freqencoder` package (`from freqencoder import FreqEncoder`)."""
from focnerf_amd.freqencoder import FreqEncoder, freq_encode  # noqa: F401
