"""Velocity-grid helpers (init-time host code; reference: Starfish/utils.py)."""
import numpy as np

from . import constants as C


def calculate_dv(wave):
    """Minimum velocity spacing of a wavelength array in km/s (Starfish/utils.py:8-22)."""
    wave = np.asarray(wave, dtype=np.float64)
    return C.c_kms * np.min(np.diff(wave) / wave[:-1])


def calculate_dv_dict(wave_dict):
    """Velocity spacing of a log-lambda ``wave_dict`` (Starfish/utils.py:25-41)."""
    return C.c_kms * (10 ** wave_dict["CDELT1"] - 1)


def create_log_lam_grid(dv, start, end):
    """Log-lambda grid with a power-of-two number of points and spacing <= dv
    (Starfish/utils.py:44-88).  Returns a dict with wl, CRVAL1, CDELT1, NAXIS1."""
    if start >= end:
        raise ValueError("Wavelength must be increasing, but start >= end")
    if start <= 0 or end <= 0:
        raise ValueError("Cannot have negative or 0 wavelength")
    step = np.log10(dv / C.c_kms + 1.0)
    crval1 = np.log10(start)
    crvaln = np.log10(end)
    want = (crvaln - crval1) / step
    naxis1 = 2
    while naxis1 < want:
        naxis1 *= 2
    cdelt1 = (crvaln - crval1) / (naxis1 - 1)
    wl = 10 ** (crval1 + cdelt1 * np.arange(naxis1))
    return {"wl": wl, "CRVAL1": crval1, "CDELT1": cdelt1, "NAXIS1": naxis1}
