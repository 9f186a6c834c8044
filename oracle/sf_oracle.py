"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A numpy/scipy restatement of the Starfish per-MCMC-step log-likelihood path, written from the
mathematics of the reference (file:line citations are relative to /root/reference).  It is the
checker for the HIP path, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; ``starfish_amd`` never does.

Pinning: every function below is checked against outputs of the real reference (imported in the
authoring container by ``tools/gen_golden.py``) through the fixtures committed under
``tests/golden/`` (``tests/test_oracle_golden.py``).  The reference's own tests hold no numeric
known-answer vectors for this path, so those generated fixtures are the pin.

Third-party arithmetic the reference delegates to and that is therefore used here as well (same
pinned dependency, ``scipy>=1.3`` / ``numpy>=1.16`` in the reference's setup.py:49-51):
LAPACK dpotrf/dpotrs/dgesv, pocketfft rfft/irfft, FITPACK curfit/splev, cephes j1.
An independent B-spline collocation restatement of the FITPACK k=5 interpolating spline
(``quintic_collocation_*``) is included because that is the algorithm the HIP kernels implement.
"""

from math import pi

import numpy as np
from numpy.polynomial.chebyshev import chebval
from scipy.interpolate import InterpolatedUnivariateSpline
from scipy.linalg import cho_factor, cho_solve
from scipy.special import j1

C_KMS = 2.99792458e5  # Starfish/constants.py:7
JITTER = 1e-10  # Starfish/models/spectrum_model.py:399


# --------------------------------------------------------------------------- grids / velocity
def min_velocity_step(wave):
    """Starfish/utils.py:8-22 -- c * min(dlam / lam)."""
    wave = np.asarray(wave, dtype=np.float64)
    return C_KMS * np.min(np.diff(wave) / wave[:-1])


def log_lambda_grid(dv, start, end):
    """Starfish/utils.py:44-88 -- power-of-two log-lambda grid with spacing <= dv."""
    if start >= end:
        raise ValueError("Wavelength must be increasing, but start >= end")
    if start <= 0 or end <= 0:
        raise ValueError("Cannot have negative or 0 wavelength")
    step = np.log10(dv / C_KMS + 1.0)
    lo = np.log10(start)
    hi = np.log10(end)
    want = (hi - lo) / step
    n = 2
    while n < want:
        n *= 2
    cdelt = (hi - lo) / (n - 1)
    return 10 ** (lo + cdelt * np.arange(n)), lo, cdelt


# --------------------------------------------------------------------------- covariance kernels
def matern32_global(wave, amplitude, lengthscale):
    """Starfish/models/kernels.py:7-41 -- Hann-tapered Matern-3/2 in velocity distance."""
    w = np.asarray(wave, dtype=np.float64)
    wi = w[None, :]
    wj = w[:, None]
    r = C_KMS / 2 * np.abs((wi - wj) / (wi + wj))
    r0 = 6 * lengthscale
    inside = r <= r0
    rr = np.where(inside, r, 0.0)
    taper = 0.5 + 0.5 * np.cos(pi * rr / r0)
    body = taper * amplitude * (1 + np.sqrt(3) * rr / lengthscale) * np.exp(
        -np.sqrt(3) * rr / lengthscale
    )
    return np.where(inside, body, 0.0)


def gaussian_local(wave, amplitude, mu, sigma):
    """Starfish/models/kernels.py:44-81 -- Hann-tapered Gaussian patch centred on mu."""
    w = np.asarray(wave, dtype=np.float64)
    d = C_KMS / mu * np.abs(w - mu)
    di = d[None, :]
    dj = d[:, None]
    r_tap = np.maximum(di, dj)
    r2 = di**2 + dj**2
    r0 = 4 * sigma
    inside = r_tap <= r0
    rt = np.where(inside, r_tap, 0.0)
    taper = 0.5 + 0.5 * np.cos(pi * rt / r0)
    body = taper * amplitude * np.exp(-0.5 * np.where(inside, r2, 0.0) / sigma**2)
    return np.where(inside, body, 0.0)


# --------------------------------------------------------------------------- transforms
def rot_broaden(wave, flux, vsini):
    """Starfish/transforms.py:93-134 -- Gray rotational kernel applied in Fourier space."""
    if vsini <= 0:
        raise ValueError("vsini must be positive")
    flux = np.asarray(flux, dtype=np.float64)
    n = flux.shape[-1]
    dv = min_velocity_step(wave)
    freq = np.fft.rfftfreq(n, dv)
    spec = np.fft.rfft(flux)
    u = (2.0 * pi * vsini * freq)[1:]
    sb = j1(u) / u - 3 * np.cos(u) / (2 * u**2) + 3.0 * np.sin(u) / (2 * u**3)
    spec *= np.concatenate(([1.0], sb))
    return np.fft.irfft(spec, n=n)


def inst_broaden(wave, flux, fwhm):
    """Starfish/transforms.py:45-90 -- Gaussian instrumental profile in Fourier space."""
    if fwhm < 0:
        raise ValueError("FWHM must be non-negative")
    flux = np.asarray(flux, dtype=np.float64)
    n = flux.shape[-1]
    dv = min_velocity_step(wave)
    freq = np.fft.rfftfreq(n, d=dv)
    spec = np.fft.rfft(flux)
    sigma = fwhm / 2.355
    spec *= np.exp(-2 * (pi * sigma * freq) ** 2)
    return np.fft.irfft(spec, n=n)


def doppler(wave, vz):
    """Starfish/transforms.py:137-158."""
    return np.asarray(wave) * np.sqrt((C_KMS + vz) / (C_KMS - vz))


def quintic_resample(wave, flux, new_wave):
    """Starfish/transforms.py:11-42 -- FITPACK interpolating k=5 spline per row."""
    new_wave = np.asarray(new_wave, dtype=np.float64)
    if np.any(new_wave <= 0):
        raise ValueError("Wavelengths must be positive")
    flux = np.asarray(flux, dtype=np.float64)
    if flux.ndim > 1:
        return np.array(
            [InterpolatedUnivariateSpline(wave, row, k=5)(new_wave) for row in flux]
        )
    return InterpolatedUnivariateSpline(wave, flux, k=5)(new_wave)


def cheb_correct(wave, flux, coeffs):
    """Starfish/transforms.py:271-304 -- multiply by a Chebyshev series in lam/lam_max."""
    coeffs = np.asarray(coeffs)
    if coeffs.ndim == 1 and coeffs[0] != 1:
        raise ValueError(
            "For single spectrum the linear Chebyshev coefficient (c[0]) must be 1"
        )
    wave = np.asarray(wave, dtype=np.float64)
    return flux * chebval(wave / wave.max(), coeffs, tensor=False)


def ccm89_a_lambda(wave, a_v, r_v=3.1):
    """A_lambda [mag] of Cardelli, Clayton & Mathis (1989, ApJ 345, 245; eqs. 2a-5b).
    PARITY UNPINNED: the reference (Starfish/transforms.py:161-206) calls the third-party
    ``extinction==0.4.*`` package (setup.py:45) which is not in /root/reference and not installed, so no
    reference output exists to pin this against; it is checked against the paper's Table 3 only."""
    x = 1e4 / np.asarray(wave, dtype=np.float64)
    a = np.zeros_like(x)
    b = np.zeros_like(x)
    ir = x < 1.1
    a[ir] = 0.574 * x[ir] ** 1.61
    b[ir] = -0.527 * x[ir] ** 1.61
    op = (x >= 1.1) & (x <= 3.3)
    y = x[op] - 1.82
    a[op] = 1 + 0.17699 * y - 0.50447 * y**2 - 0.02427 * y**3 + 0.72085 * y**4 + 0.01979 * y**5 \
        - 0.77530 * y**6 + 0.32999 * y**7
    b[op] = 1.41338 * y + 2.28305 * y**2 + 1.07233 * y**3 - 5.38434 * y**4 - 0.62251 * y**5 \
        + 5.30260 * y**6 - 2.09002 * y**7
    uv = (x > 3.3) & (x <= 8.0)
    xu = x[uv]
    d = np.where(xu >= 5.9, xu - 5.9, 0.0)
    a[uv] = 1.752 - 0.316 * xu - 0.104 / ((xu - 4.67) ** 2 + 0.341) - 0.04473 * d**2 - 0.009779 * d**3
    b[uv] = -3.090 + 1.825 * xu + 1.206 / ((xu - 4.62) ** 2 + 0.263) + 0.2130 * d**2 + 0.1207 * d**3
    fuv = x > 8.0
    d = x[fuv] - 8.0
    a[fuv] = -1.073 - 0.628 * d + 0.137 * d**2 - 0.070 * d**3
    b[fuv] = 13.670 + 4.257 * d - 0.420 * d**2 + 0.374 * d**3
    return a_v * (a + b / r_v)


def odonnell94_a_lambda(wave, a_v, r_v=3.1):
    """O'Donnell (1994, ApJ 422, 158): CCM89 with re-derived a(y), b(y) for 1.1 <= x <= 3.3 um^-1.
    PARITY UNPINNED (see ccm89_a_lambda)."""
    w = np.asarray(wave, dtype=np.float64)
    x = 1e4 / w
    out = ccm89_a_lambda(w, a_v, r_v)
    op = (x >= 1.1) & (x <= 3.3)
    y = x[op] - 1.82
    a = 1 + 0.104 * y - 0.609 * y**2 + 0.701 * y**3 + 1.137 * y**4 - 1.718 * y**5 - 0.827 * y**6 \
        + 1.647 * y**7 - 0.505 * y**8
    b = 1.952 * y + 2.908 * y**2 - 3.989 * y**3 - 7.985 * y**4 + 11.102 * y**5 + 5.491 * y**6 \
        - 10.805 * y**7 + 3.347 * y**8
    out = np.array(out, dtype=np.float64)
    out[op] = a_v * (a + b / r_v)
    return out


def calzetti00_a_lambda(wave, a_v, r_v=4.05):
    """Calzetti et al. (2000, ApJ 533, 682) eq. 4; A_lambda = Av k(lambda) / Rv.  PARITY UNPINNED."""
    l = np.asarray(wave, dtype=np.float64) * 1e-4
    k = np.where(l >= 0.63, 2.659 * (-1.857 + 1.040 / l) + r_v,
                 2.659 * (-2.156 + 1.509 / l - 0.198 / l**2 + 0.011 / l**3) + r_v)
    return a_v * k / r_v


def _fm_uv(x, c1, c2, c3, c4, c5, x0, gamma, f99):
    x = np.asarray(x, dtype=np.float64)
    k = c1 + c2 * x + c3 * x**2 / ((x**2 - x0**2) ** 2 + x**2 * gamma**2)
    y = np.where(x >= c5, x - c5, 0.0)
    return k + (c4 * (0.5392 * y**2 + 0.05644 * y**3) if f99 else c4 * y**2)


def fitzpatrick99_a_lambda(wave, a_v, r_v=3.1):
    """Fitzpatrick (1999, PASP 111, 63): natural cubic spline in x = 1/lambda through Rv-dependent anchor points
    (his FM_UNRED coding) below 1e4/2700 um^-1, FM90 ultraviolet curve above.  PARITY UNPINNED (see ccm89_a_lambda)."""
    from scipy.interpolate import CubicSpline

    c2 = -0.824 + 4.717 / r_v
    c1 = 2.030 - 3.007 * c2
    uv = (c1, c2, 3.23, 0.41, 5.9, 4.596, 0.99, True)
    xk = np.array([0.0, 1e4 / 26500, 1e4 / 12200, 1e4 / 6000, 1e4 / 5470, 1e4 / 4670, 1e4 / 4110, 1e4 / 2700, 1e4 / 2600])
    yk = np.array([
        -r_v, 0.26469 * r_v / 3.1 - r_v, 0.82925 * r_v / 3.1 - r_v,
        -4.22809e-01 + 1.00270 * r_v + 2.13572e-04 * r_v**2 - r_v,
        -5.13540e-02 + 1.00216 * r_v - 7.35778e-05 * r_v**2 - r_v,
        7.00127e-01 + 1.00184 * r_v - 3.32598e-05 * r_v**2 - r_v,
        1.19456 + 1.01707 * r_v - 5.46959e-03 * r_v**2 + 7.97809e-04 * r_v**3 - 4.45636e-05 * r_v**4 - r_v,
        *_fm_uv(xk[-2:], *uv),
    ])
    x = 1e4 / np.asarray(wave, dtype=np.float64)
    k = np.where(x >= xk[-2], _fm_uv(x, *uv), CubicSpline(xk, yk, bc_type="natural")(np.minimum(x, xk[-1])))
    return a_v * (1 + k / r_v)


def fm07_a_lambda(wave, a_v):
    """Fitzpatrick & Massa (2007, ApJ 663, 320) mean curve, Rv = 3.1.  PARITY UNPINNED."""
    from scipy.interpolate import CubicSpline

    r_v = 3.1
    uv = (-0.175, 0.807, 2.991, 0.319, 6.097, 4.592, 0.922, False)
    xk = np.array([0.0, 0.25, 0.50, 0.75, 1.0, 1e4 / 5530, 1e4 / 4000, 1e4 / 3300, 1e4 / 2700, 1e4 / 2600])
    yk = np.concatenate([(-0.83 + 0.63 * r_v) * xk[:5] ** 1.84 - r_v, [0.0, 1.322, 2.055], _fm_uv(xk[-2:], *uv)])
    x = 1e4 / np.asarray(wave, dtype=np.float64)
    k = np.where(x >= xk[-2], _fm_uv(x, *uv), CubicSpline(xk, yk, bc_type="natural")(np.minimum(x, xk[-1])))
    return a_v * (1 + k / r_v)


def extinct_ccm89(wave, flux, a_v, r_v=3.1):
    """flux * 10**(-0.4 A_lambda)  (Starfish/transforms.py:205).  PARITY UNPINNED, see above."""
    return flux * 10 ** (-0.4 * ccm89_a_lambda(wave, a_v, r_v))


def trapezoid(y, x):
    """numpy.trapz as called from Starfish/transforms.py:265-268."""
    y = np.asarray(y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    return (np.diff(x) * (y[..., 1:] + y[..., :-1]) / 2.0).sum(axis=-1)


def renorm_factor(wave, flux, reference_flux):
    """Starfish/transforms.py:265-268."""
    return trapezoid(reference_flux, wave) / trapezoid(flux, wave)


# ------------------------------------------------- B-spline collocation restatement of FITPACK k=5
def quintic_knots(x):
    """Knot vector FITPACK's curfit uses for an interpolating (s=0) k=5 spline:
    x[0] six times, x[3:-3], x[-1] six times (fpcurf.f, 'iopt=0, s=0' branch)."""
    x = np.asarray(x, dtype=np.float64)
    return np.concatenate((np.repeat(x[0], 6), x[3:-3], np.repeat(x[-1], 6)))


def bspline_basis6(t, ell, x):
    """The six degree-5 B-splines that are non-zero on [t[ell], t[ell+1]) evaluated at x,
    by the Cox-de Boor recurrence exactly as FITPACK's fpbspl.f orders it.
    Returns h[0..5] multiplying coefficients c[ell-5 .. ell]."""
    h = np.zeros(6)
    hh = np.zeros(5)
    h[0] = 1.0
    for j in range(1, 6):
        hh[:j] = h[:j]
        h[0] = 0.0
        for i in range(1, j + 1):
            li = ell + i
            lj = li - j
            f = hh[i - 1] / (t[li] - t[lj])
            h[i - 1] = h[i - 1] + f * (t[li] - x)
            h[i] = f * (x - t[lj])
    return h


def find_interval(t, n_coef, x):
    """splev.f: largest ell with t[ell] <= x, clamped to [5, n_coef-1] (0-based)."""
    ell = int(np.searchsorted(t, x, side="right")) - 1
    return min(max(ell, 5), n_coef - 1)


def quintic_collocation_band(x):
    """Banded collocation matrix A[i, j] = B_j(x_i) for the interpolation problem.
    Returns (ab, offs): ab[i, :] holds A[i, offs[i] : offs[i]+6]."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    t = quintic_knots(x)
    ab = np.zeros((n, 6))
    offs = np.zeros(n, dtype=np.int64)
    for i in range(n):
        ell = find_interval(t, n, x[i])
        ab[i] = bspline_basis6(t, ell, x[i])
        offs[i] = ell - 5
    return ab, offs, t


def quintic_collocation_fit(x, y):
    """Dense-solve version of the interpolation conditions (small n only; test helper)."""
    ab, offs, t = quintic_collocation_band(x)
    n = len(x)
    A = np.zeros((n, n))
    for i in range(n):
        A[i, offs[i] : offs[i] + 6] = ab[i]
    return np.linalg.solve(A, np.asarray(y, dtype=np.float64)), t


def quintic_collocation_eval(t, c, xq):
    out = np.empty(len(xq))
    n = len(c)
    for q, x in enumerate(xq):
        ell = find_interval(t, n, x)
        out[q] = bspline_basis6(t, ell, x) @ c[ell - 5 : ell + 1]
    return out


# --------------------------------------------------------------------------- emulator (query side)
def rbf_block(X, Z, variance, lengthscale):
    """Starfish/emulator/kernels.py:5-26 -- sigma^2 exp(-1/2 |(x-z)/l|^2)."""
    X = np.atleast_2d(X) / lengthscale
    Z = np.atleast_2d(Z) / lengthscale
    d2 = ((X[:, None, :] - Z[None, :, :]) ** 2).sum(-1)
    return variance * np.exp(-0.5 * d2)


def rbf_blockdiag(X, Z, variances, lengthscales):
    """Starfish/emulator/kernels.py:29-49 -- block-diagonal stack, one RBF block per component."""
    blocks = [rbf_block(X, Z, v, l) for v, l in zip(variances, lengthscales)]
    rows = sum(b.shape[0] for b in blocks)
    cols = sum(b.shape[1] for b in blocks)
    out = np.zeros((rows, cols))
    r = c = 0
    for b in blocks:
        out[r : r + b.shape[0], c : c + b.shape[1]] = b
        r += b.shape[0]
        c += b.shape[1]
    return out


def phi_squared(eigenspectra, M):
    """Starfish/emulator/_utils.py:28-48 -- Phi^T Phi = (E E^T) kron I_M (component-major)."""
    E = np.asarray(eigenspectra, dtype=np.float64)
    dots = E @ E.T
    return np.kron(dots, np.eye(M))


def w_hat_estimate(eigenspectra, fluxes):
    """Starfish/emulator/_utils.py:10-25 -- least-squares PCA weights, component-major."""
    E = np.asarray(eigenspectra, dtype=np.float64)
    F = np.asarray(fluxes, dtype=np.float64)
    M = len(F)
    rhs = (E @ F.T).reshape(-1)
    fac = cho_factor(phi_squared(E, M))
    return cho_solve(fac, rhs)


def emulator_v11(eigenspectra, grid_points, variances, lengthscales, lambda_xi=1.0):
    """Starfish/emulator/emulator.py:123-128."""
    M = len(grid_points)
    return np.linalg.inv(phi_squared(eigenspectra, M)) / lambda_xi + rbf_blockdiag(
        grid_points, grid_points, variances, lengthscales
    )


def default_lengthscales(grid_points, ncomps):
    """Starfish/emulator/emulator.py:109-116 -- 3 x the largest grid step per axis."""
    g = np.asarray(grid_points, dtype=np.float64)
    sep = np.array([np.diff(np.unique(col)).max() for col in g.T])
    return np.tile(3 * sep, (ncomps, 1))


def emulator_query(grid_points, params, variances, lengthscales, v11, w_hat):
    """Starfish/emulator/emulator.py:330-394 -- GP conditional mean / covariance of the weights."""
    params = np.atleast_2d(params)
    g = np.asarray(grid_points, dtype=np.float64)
    if np.any(params < g.min(axis=0)) or np.any(params > g.max(axis=0)):
        raise ValueError("Querying emulator outside of original parameter range.")
    v12 = rbf_blockdiag(g, params, variances, lengthscales)
    v22 = rbf_blockdiag(params, params, variances, lengthscales)
    mu = v12.T @ np.linalg.solve(v11, w_hat)
    cov = v22 - v12.T @ np.linalg.solve(v11, v12)
    return mu, cov


# --------------------------------------------------------------------------- the model
class OracleOrder:
    """Static per-order state mirroring SpectrumModel.__init__
    (Starfish/models/spectrum_model.py:126-181) for a single order."""

    def __init__(
        self,
        wave,
        flux,
        sigma,
        emu_wl,
        eigenspectra,
        flux_mean,
        flux_std,
        grid_points,
        w_hat,
        variances=None,
        lengthscales=None,
        lambda_xi=1.0,
    ):
        self.wave = np.asarray(wave, dtype=np.float64)
        self.flux = np.asarray(flux, dtype=np.float64)
        self.sigma = np.asarray(sigma, dtype=np.float64)
        self.emu_wl = np.asarray(emu_wl, dtype=np.float64)
        self.eigenspectra = np.asarray(eigenspectra, dtype=np.float64)
        self.m = self.eigenspectra.shape[0]
        self.grid_points = np.asarray(grid_points, dtype=np.float64)
        self.w_hat = np.asarray(w_hat, dtype=np.float64)
        self.variances = (
            np.asarray(variances, dtype=np.float64)
            if variances is not None
            else 1e4 * np.ones(self.m)
        )
        self.lengthscales = (
            np.asarray(lengthscales, dtype=np.float64)
            if lengthscales is not None
            else default_lengthscales(self.grid_points, self.m)
        )
        self.v11 = emulator_v11(
            self.eigenspectra, self.grid_points, self.variances, self.lengthscales, lambda_xi
        )
        bulk = np.vstack([self.eigenspectra, flux_mean, flux_std])  # emulator.py:396-402
        dv = min_velocity_step(self.wave)
        self.min_dv_wave, _, _ = log_lambda_grid(dv, self.emu_wl.min(), self.emu_wl.max())
        self.bulk_fluxes = quintic_resample(self.emu_wl, bulk, self.min_dv_wave)


def forward_model(order, p):
    """Steps 1-10 of SURVEY.md section 0 == SpectrumModel.__call__
    (Starfish/models/spectrum_model.py:277-365).

    ``p`` is a plain dict: optional vsini, vz, cheb (c1..), log_scale, norm (float factor),
    global_cov=(log_amp, log_ls), local_cov=[(mu, log_amp, log_sigma), ...], grid=[...].
    Returns flux (N,), cov (N, N) and the scale factor used."""
    wave = order.min_dv_wave
    rows = order.bulk_fluxes
    if "vsini" in p:
        rows = rot_broaden(wave, rows, p["vsini"])
    if "vz" in p:
        wave = doppler(wave, p["vz"])
    rows = quintic_resample(wave, rows, order.wave)
    if "Av" in p:  # spectrum_model.py:298-299 (Rv is never passed: 3.1); PARITY UNPINNED
        rows = extinct_ccm89(order.wave, rows, p["Av"])
    if "cheb" in p:
        rows = cheb_correct(order.wave, rows, [1, *p["cheb"]])
    w_mu, w_cov = emulator_query(
        order.grid_points, p["grid"], order.variances, order.lengthscales, order.v11, order.w_hat
    )
    eig, mean, std = rows[:-2], rows[-2], rows[-1]
    X = eig * std
    flux = w_mu @ X + mean
    norm = p.get("norm", 1)
    if "log_scale" in p:
        scale = np.exp(p["log_scale"]) * norm
    else:
        scale = renorm_factor(order.wave, flux * norm, order.flux) * norm
    flux = flux * scale
    X = X * scale
    if p.get("emulator_cov", "code") == "paper":
        # the form printed in the paper / docs (docs/api/emulator.rst:105): Phi Sigma_w Phi^T; non-default switch
        cov = X.T @ w_cov @ X
    else:
        fac = cho_factor(w_cov)
        cov = X.T @ cho_solve(fac, X)  # what the code does: spectrum_model.py:334-335
    idx = np.arange(len(order.wave))
    cov[idx, idx] += order.sigma**2
    if "global_cov" in p:
        la, ll = p["global_cov"]
        cov += matern32_global(order.wave, np.exp(la), np.exp(ll))
    if p.get("local_cov"):
        loc = 0
        for mu, la, ls in p["local_cov"]:
            loc = loc + gaussian_local(order.wave, np.exp(la), mu, np.exp(ls))
        cov += loc
    return flux, cov, scale


def log_likelihood(order, p, return_parts=False):
    """SpectrumModel.log_likelihood without priors
    (Starfish/models/spectrum_model.py:397-405): -(logdet + R^T C^-1 R) / 2."""
    flux, cov, _ = forward_model(order, p)
    idx = np.arange(len(flux))
    cov[idx, idx] += JITTER
    fac = cho_factor(cov, overwrite_a=True)
    logdet = 2 * np.sum(np.log(fac[0].diagonal()))
    R = flux - order.flux
    sqmah = R @ cho_solve(fac, R)
    lnl = -(logdet + sqmah) / 2
    if return_parts:
        return lnl, logdet, sqmah, R
    return lnl
