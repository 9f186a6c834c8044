"""
Synthetic single-order problems: plain numpy arrays only, so the same
inputs can be handed to the HIP path, to the CPU oracle and -- in the authoring container -- to the
real reference (tools/gen_golden.py).  All randomness is ``np.random.default_rng(seed)``.
"""

from itertools import product

import numpy as np

C_KMS = 2.99792458e5


def _log_grid(dv, start, end):
    step = np.log10(dv / C_KMS + 1.0)
    lo, hi = np.log10(start), np.log10(end)
    n = 2
    while n < (hi - lo) / step:
        n *= 2
    return 10 ** (lo + (hi - lo) / (n - 1) * np.arange(n))


def make_order(N=4096, m=8, seed=0, dv=2.0, wave0=5000.0, pad=20.0, grid_axes=None):
    """One echelle order of N pixels and a matching m-component emulator over a 3x3x3 library grid
    (``grid_axes``: other axis values, e.g. the 11 x 6 x 5 = 330 points of BIG_GRID_AXES)."""
    rng = np.random.default_rng(seed)
    wave = wave0 * np.exp(np.arange(N) * dv / C_KMS)
    emu_wl = _log_grid(dv, wave.min() - pad, wave.max() + pad)
    nf = len(emu_wl)
    q, _ = np.linalg.qr(rng.standard_normal((nf, m)))
    eig = np.ascontiguousarray(q.T)
    flux_mean = 1 + 0.1 * np.sin(emu_wl / 7)
    flux_std = 0.05 + 0.01 * np.cos(emu_wl / 3)
    axes = grid_axes or ((6000.0, 6100.0, 6200.0), (4.0, 4.5, 5.0), (-1.0, -0.5, 0.0))
    grid = np.array(list(product(*axes)))
    M = len(grid)
    weights = rng.standard_normal((M, m))
    fluxes = weights @ eig
    # least-squares PCA weights, component-major (i*M + j)
    dots = eig @ eig.T
    rhs = (eig @ fluxes.T).reshape(-1)
    w_hat = np.linalg.solve(np.kron(dots, np.eye(M)), rhs)
    data_flux = 1 + 0.1 * np.sin(wave / 7) + 0.01 * rng.standard_normal(N)
    sigma = 0.01 * np.ones(N)
    return dict(
        wave=wave,
        flux=data_flux,
        sigma=sigma,
        emu_wl=emu_wl,
        eigenspectra=eig,
        flux_mean=flux_mean,
        flux_std=flux_std,
        grid_points=grid,
        param_names=["T", "logg", "Z"],
        weights=weights,
        w_hat=w_hat,
        factors=np.ones(M),
    )


# a library of the size of the reference's worked example (examples/setup.ipynb:47,185: m = 4, M = 330 -> m M = 1320)
BIG_GRID_AXES = (
    tuple(5700.0 + 100.0 * i for i in range(11)),
    tuple(3.5 + 0.5 * i for i in range(6)),
    tuple(-2.0 + 0.5 * i for i in range(5)),
)


def centre_params(order):
    """The centre of the walker ball."""
    wave = order["wave"]
    N = len(wave)
    return dict(
        vz=10.0,
        vsini=30.0,
        log_scale=0.0,
        global_cov=dict(log_amp=-9.0, log_ls=float(np.log(10.0))),
        local_cov=[
            dict(mu=float(wave[N // 3]), log_amp=-8.0, log_sigma=float(np.log(15.0)))
        ],
        cheb=[0.01, -0.02],
        grid_params=[6050.0, 4.2, -0.3],
    )


# label order produced by SpectrumModel for centre_params (kwargs order, cheb last, then grid)
LABELS = (
    "vz",
    "vsini",
    "log_scale",
    "global_cov:log_amp",
    "global_cov:log_ls",
    "local_cov:0:mu",
    "local_cov:0:log_amp",
    "local_cov:0:log_sigma",
    "cheb:1",
    "cheb:2",
    "T",
    "logg",
    "Z",
)

_BALL = {
    "vz": 0.1,
    "vsini": 0.1,
    "log_scale": 0.01,
    "global_cov:log_amp": 0.05,
    "global_cov:log_ls": 0.05,
    "local_cov:0:mu": 0.01,
    "local_cov:0:log_amp": 0.05,
    "local_cov:0:log_sigma": 0.05,
    "cheb:1": 1e-3,
    "cheb:2": 1e-3,
    "T": 1.0,
    "logg": 0.01,
    "Z": 0.01,
}


def centre_vector(order):
    c = centre_params(order)
    return np.array(
        [
            c["vz"],
            c["vsini"],
            c["log_scale"],
            c["global_cov"]["log_amp"],
            c["global_cov"]["log_ls"],
            c["local_cov"][0]["mu"],
            c["local_cov"][0]["log_amp"],
            c["local_cov"][0]["log_sigma"],
            c["cheb"][0],
            c["cheb"][1],
            *c["grid_params"],
        ]
    )


def walker_ball(order, B=128, seed=1):
    """B parameter vectors (B, 13) in LABELS order scattered around the centre."""
    rng = np.random.default_rng(seed)
    p0 = centre_vector(order)
    scales = np.array([_BALL[k] for k in LABELS])
    return p0[None, :] + scales[None, :] * rng.standard_normal((B, len(LABELS)))


def vector_to_oracle_params(vec):
    """LABELS-ordered vector -> the dict understood by oracle.sf_oracle.forward_model."""
    v = dict(zip(LABELS, vec))
    return dict(
        vz=v["vz"],
        vsini=v["vsini"],
        log_scale=v["log_scale"],
        global_cov=(v["global_cov:log_amp"], v["global_cov:log_ls"]),
        local_cov=[(v["local_cov:0:mu"], v["local_cov:0:log_amp"], v["local_cov:0:log_sigma"])],
        cheb=[v["cheb:1"], v["cheb:2"]],
        grid=[v["T"], v["logg"], v["Z"]],
    )


# ---------------------------------------------------------------------------------------------
# Multi-order problems (cfg 3/4): independent orders, each a cfg-2 style order
# with its own emulator chunk; stellar / calibration parameters are shared, every order keeps its own
# (frozen) local kernel.
SHARED_LABELS = tuple(k for k in LABELS if not k.startswith("local_cov"))


def make_echelle(n_orders=25, N=3000, m=8, seed0=100):
    """``n_orders`` synthetic orders starting at 5000 * 1.02**o Angstrom."""
    return [make_order(N=N, m=m, seed=seed0 + o, wave0=5000.0 * 1.02**o) for o in range(n_orders)]


def shared_ball(order0, B=64, seed=1):
    """B shared parameter vectors (B, 10) in SHARED_LABELS order around the centre."""
    keep = [i for i, k in enumerate(LABELS) if not k.startswith("local_cov")]
    return walker_ball(order0, B=B, seed=seed)[:, keep]


def shared_to_oracle_params(order, vec):
    """SHARED_LABELS vector + the order's own fixed local kernel -> oracle parameter dict."""
    c = centre_params(order)
    v = dict(zip(SHARED_LABELS, vec))
    return dict(
        vz=v["vz"],
        vsini=v["vsini"],
        log_scale=v["log_scale"],
        global_cov=(v["global_cov:log_amp"], v["global_cov:log_ls"]),
        local_cov=[(k["mu"], k["log_amp"], k["log_sigma"]) for k in c["local_cov"]],
        cheb=[v["cheb:1"], v["cheb:2"]],
        grid=[v["T"], v["logg"], v["Z"]],
    )


def perturb_grid(order, rng_seed=7):
    """The same order on a wavelength grid that is NOT log-uniform (pixel spacing modulated by +-5 %, like a real
    rectified order): the likelihood path then evaluates K_global per entry instead of from the per-diagonal table."""
    w = order["wave"]
    n = len(w)
    step = np.diff(w) * (1 + 0.05 * np.sin(np.arange(n - 1) / 37.0))
    out = dict(order)
    out["wave"] = np.concatenate([[w[0]], w[0] + np.cumsum(step)])
    out["flux"] = 1 + 0.1 * np.sin(out["wave"] / 7) + 0.01 * np.random.default_rng(rng_seed).standard_normal(n)
    return out


def build_model(order, params=None, device=None, solver="dense", freeze=(), emulator_cov="code"):
    """A product ``SpectrumModel`` (own ``Emulator`` + ``Spectrum``) for a synthetic order: the path a user
    takes, including the model's own init-time resample of the emulator's bulk fluxes."""
    from . import Spectrum
    from .emulator import Emulator
    from .models import SpectrumModel

    emu = Emulator(order["grid_points"], order["param_names"], order["emu_wl"], order["weights"],
                   order["eigenspectra"], order["w_hat"], order["flux_mean"], order["flux_std"], order["factors"])
    emu._trained = True
    data = Spectrum(order["wave"], order["flux"], sigmas=order["sigma"])
    c = dict(centre_params(order)) if params is None else dict(params)
    gp = c.pop("grid_params")
    model = SpectrumModel(emu, data, grid_params=gp, device=device, solver=solver, emulator_cov=emulator_cov, **c)
    for name in freeze:
        model.freeze(name)
    return model


def build_echelle(orders, device=None, devices=None):
    """Multi-order model over synthetic orders: shared stellar / calibration parameters, every order keeps its
    own frozen local kernel (labels == SHARED_LABELS)."""
    from .models import EchelleModel

    models = []
    for i, o in enumerate(orders):
        dev = device if devices is None else devices[i % len(devices)]
        models.append(build_model(o, device=dev, freeze=("local_cov",)))
    return EchelleModel.from_orders(models)
