"""RBF kernels of the emulator GP (reference: Starfish/emulator/kernels.py).  These host versions
only build the constant ``v11`` at construction / hyper-parameter changes; the per-step ``v12``
blocks are evaluated inside the ``k_emu_prep`` HIP kernel."""
import numpy as np


def rbf_kernel(X, Z, variance, lengthscale):
    """variance * exp(-1/2 |(x - z) / lengthscale|^2)  (Starfish/emulator/kernels.py:5-26)."""
    Xs = np.atleast_2d(X) / lengthscale
    Zs = np.atleast_2d(Z) / lengthscale
    d2 = ((Xs[:, None, :] - Zs[None, :, :]) ** 2).sum(axis=-1)
    return variance * np.exp(-0.5 * d2)


def batch_kernel(X, Z, variances, lengthscales):
    """Block-diagonal stack of one RBF block per component (Starfish/emulator/kernels.py:29-49)."""
    blocks = [rbf_kernel(X, Z, v, l) for v, l in zip(variances, lengthscales)]
    nr = sum(b.shape[0] for b in blocks)
    nc = sum(b.shape[1] for b in blocks)
    out = np.zeros((nr, nc))
    r = c = 0
    for b in blocks:
        out[r : r + b.shape[0], c : c + b.shape[1]] = b
        r += b.shape[0]
        c += b.shape[1]
    return out
