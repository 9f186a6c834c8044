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
