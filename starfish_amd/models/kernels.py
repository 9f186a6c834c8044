"""
Covariance kernel builders (reference: Starfish/models/kernels.py), evaluated by the gfx950 HIP
kernels ``k_global_cov`` / ``k_local_cov``.  Inside ``SpectrumModel`` both are fused into the single
covariance fill pass (``k_fill``); these free functions exist for API parity and stage tests.
"""
import numpy as np

from .. import _device as D
from .. import _lib


def global_covariance_matrix(wave, amplitude, lengthscale):
    """Hann-tapered Matern-3/2 kernel in velocity distance (Starfish/models/kernels.py:7-41)."""
    lib = _lib.require_gpu()
    wave = np.asarray(wave, dtype=np.float64)
    dev = D.device_of()
    n = wave.shape[0]
    d_wave = D.to_dev(wave, dev)
    d_out = D.empty((n, n), dev)
    rc = lib.sf_global_cov(D.ptr(d_wave), n, float(amplitude), float(lengthscale), D.ptr(d_out),
                           D.stream_ptr(dev))
    _lib.check(rc, "sf_global_cov")
    return d_out.cpu().numpy()


def local_covariance_matrix(wave, amplitude, mu, sigma):
    """Hann-tapered Gaussian patch centred on ``mu`` (Starfish/models/kernels.py:44-81)."""
    lib = _lib.require_gpu()
    wave = np.asarray(wave, dtype=np.float64)
    dev = D.device_of()
    n = wave.shape[0]
    d_wave = D.to_dev(wave, dev)
    d_out = D.empty((n, n), dev)
    rc = lib.sf_local_cov(D.ptr(d_wave), n, float(amplitude), float(mu), float(sigma), 0, D.ptr(d_out),
                          D.stream_ptr(dev))
    _lib.check(rc, "sf_local_cov")
    return d_out.cpu().numpy()
