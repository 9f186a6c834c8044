"""
starfish_amd -- MI355X (gfx950) implementation of the Starfish per-MCMC-step log-likelihood path
behind the reference's own Python API (``SpectrumModel`` / ``Emulator`` / ``Spectrum`` and the free
functions of ``transforms`` and ``models.kernels``).  All numerics run in hand-written HIP kernels
reached through the C-ABI in ``include/starfish_amd.h``; there is no CPU fallback.
"""

__version__ = "0.1.0"

from .spectrum import Order, Spectrum  # noqa: E402,F401

__all__ = ["constants", "emulator", "models", "spectrum", "Spectrum", "Order", "transforms", "utils"]
