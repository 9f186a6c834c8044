"""Init-time linear-algebra helpers of the emulator (reference: Starfish/emulator/_utils.py)."""
import numpy as np


def get_phi_squared(eigenspectra, M):
    """Phi^T Phi without forming Phi: <e_i, e_j> on the (i*M + k, j*M + k) entries, i.e.
    (E E^T) kron I_M in the component-major ordering (Starfish/emulator/_utils.py:28-48)."""
    E = np.asarray(eigenspectra, dtype=np.float64)
    return np.kron(E @ E.T, np.eye(M))


def get_w_hat(eigenspectra, fluxes):
    """Least-squares PCA weights of the library spectra, component-major
    (Starfish/emulator/_utils.py:10-25).  Set-up only; not on the per-step path."""
    E = np.asarray(eigenspectra, dtype=np.float64)
    F = np.asarray(fluxes, dtype=np.float64)
    rhs = (E @ F.T).reshape(-1)
    return np.linalg.solve(get_phi_squared(E, len(F)), rhs)


def get_altered_prior_factors(eigenspectra, fluxes):
    """Shape and rate of the Gamma prior on lambda_xi after the reconstruction error of the library has been
    absorbed (Czekala et al. 2015, eqs. A24-A25; Starfish/emulator/_utils.py:51-82):
    a' = M (N_pix - m) / 2,   b' = (F.F - F.(Phi w_hat)) / 2."""
    E = np.asarray(eigenspectra, dtype=np.float64)
    F = np.asarray(fluxes, dtype=np.float64)
    M, npix = F.shape
    m = len(E)
    recon = get_w_hat(E, F).reshape(m, M).T @ E  # row i: sum_j w_hat[j M + i] e_j
    return 0.5 * M * (npix - m), 0.5 * float(np.sum(F * F) - np.sum(F * recon))


class Gamma:
    """Gamma(alpha, rate beta) density, the prior family of the emulator's lambda_xi
    (Starfish/emulator/_utils.py:85-101)."""

    def __init__(self, alpha, beta=1):
        self.alpha = alpha
        self.beta = beta

    def logpdf(self, x):
        from scipy.special import gammaln

        x = np.asarray(x, dtype=np.float64)
        return self.alpha * np.log(self.beta) - gammaln(self.alpha) + (self.alpha - 1) * np.log(x) - self.beta * x

    def pdf(self, x):
        return np.exp(self.logpdf(x))
