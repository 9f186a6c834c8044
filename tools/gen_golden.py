"""
Generate the golden fixtures under tests/golden/ by importing the REAL reference
(/root/reference, read-only) in the authoring container.  The reference never travels: only the
input/output vectors written here are committed.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [--big]

Import-time stand-ins for packages missing from this image live in tools/ref_stubs/ (they are this
repo's own files; `flatdict` re-exports starfish_amd._flatdict.FlatterDict, the others are never
called).  ``--big`` additionally produces the N=4096 batch and the N=16384 single-walker vectors
(minutes of CPU and ~25 GB of RAM).
"""

import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "ref_stubs"), "/root/reference", ROOT]
warnings.simplefilter("ignore")

import numpy as np  # noqa: E402

from Starfish import Spectrum  # noqa: E402
from Starfish.emulator import Emulator  # noqa: E402
from Starfish.models import SpectrumModel  # noqa: E402
from Starfish.models.kernels import (  # noqa: E402
    global_covariance_matrix,
    local_covariance_matrix,
)
from Starfish import transforms as T  # noqa: E402
from Starfish.utils import calculate_dv, create_log_lam_grid  # noqa: E402

from starfish_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def ref_objects(o, variances=None, lengthscales=None, factors=None):
    emu = Emulator(
        o["grid_points"],
        o["param_names"],
        o["emu_wl"],
        o["weights"],
        o["eigenspectra"],
        o["w_hat"],
        o["flux_mean"],
        o["flux_std"],
        o["factors"] if factors is None else factors,
        variances=variances,
        lengthscales=lengthscales,
    )
    emu._trained = True
    data = Spectrum(o["wave"], o["flux"], sigmas=o["sigma"])
    return emu, data


def ref_model(o, params=None, norm=False, **kw):
    emu, data = ref_objects(o, **kw)
    c = dict(synth.centre_params(o)) if params is None else dict(params)
    gp = c.pop("grid_params")
    return SpectrumModel(emu, data, grid_params=gp, norm=norm, **c)


def parts(model):
    """lnL with its pieces, recomputed the way log_likelihood does it."""
    from scipy.linalg import cho_factor, cho_solve

    ll = model.log_likelihood()
    flux, cov = model()
    cov_j = cov.copy()
    np.fill_diagonal(cov_j, cov_j.diagonal() + 1e-10)
    fac = cho_factor(cov_j)
    logdet = 2 * np.sum(np.log(fac[0].diagonal()))
    R = flux - model.data.flux
    sqmah = R @ cho_solve(fac, R)
    assert abs(-(logdet + sqmah) / 2 - ll) <= 1e-9 * abs(ll)
    return ll, logdet, sqmah, flux, cov


# ------------------------------------------------------------------------------------ kernels
def gen_kernels():
    out = {}
    for N in (64, 200):
        wave = synth.make_order(N=N, m=2)["wave"]
        out[f"wave_{N}"] = wave
        dv = calculate_dv(wave)
        r_pair = 2.99792458e5 / 2 * abs(wave[10] - wave[3]) / (wave[10] + wave[3])
        g_cases = [
            (np.exp(-9.0), 10.0),
            (2.5, 1.0),
            (0.3, 200.0),
            (1.7e-3, r_pair / 6.0),  # r0 lands (to rounding) on an actual pixel-pair distance
            (1.0, 0.2 * dv),  # band narrower than one pixel: diagonal only
        ]
        l_cases = [
            (np.exp(-8.0), wave[N // 3], 15.0),
            (0.7, wave[1], 10.0),  # patch clipped by the array edge
            (0.2, wave[0] - 0.3, 25.0),  # centre outside the array
            (3.0, wave[N // 2] + 0.004, 0.5),  # narrower than a pixel or two
            (1e-3, wave[-1], 400.0),  # patch covering everything
        ]
        out[f"g_params_{N}"] = np.array(g_cases)
        out[f"l_params_{N}"] = np.array(l_cases)
        for i, (a, l) in enumerate(g_cases):
            out[f"g_{N}_{i}"] = global_covariance_matrix(wave, a, l)
        for i, (a, mu, s) in enumerate(l_cases):
            out[f"l_{N}_{i}"] = local_covariance_matrix(wave, a, mu, s)
    np.savez_compressed(os.path.join(OUT, "kernels.npz"), **out)
    print("kernels.npz", len(out))


# --------------------------------------------------------------------------------- transforms
def gen_transforms():
    out = {}
    rng = np.random.default_rng(7)
    for tag, nf, ndata in (("s", 512, 300), ("l", 8192, 4096)):
        dv = 2.0
        wave = 5000 * np.exp(np.arange(ndata) * dv / 2.99792458e5)
        grid = create_log_lam_grid(dv, wave.min() - 2, wave.max() + 2)["wl"]
        if len(grid) != nf:  # keep the intended FFT length
            grid = create_log_lam_grid(dv, wave.min() - 20, wave.max() + 20)["wl"]
        assert len(grid) == nf, (len(grid), nf)
        rows = 3 if tag == "s" else 2
        flux = 1 + 0.1 * np.sin(grid / 3)[None, :] + 0.05 * rng.standard_normal((rows, nf))
        out[f"{tag}_grid"] = grid
        out[f"{tag}_wave"] = wave
        out[f"{tag}_flux"] = flux
        vs = (0.5, 30.0, 300.0) if tag == "s" else (30.0,)
        out[f"{tag}_vsini"] = np.array(vs)
        for i, v in enumerate(vs):
            out[f"{tag}_rot_{i}"] = T.rotational_broaden(grid, flux, v)
        fw = (0.0, 6.8, 400.0) if tag == "s" else (6.8,)
        out[f"{tag}_fwhm"] = np.array(fw)
        for i, f in enumerate(fw):
            out[f"{tag}_inst_{i}"] = T.instrumental_broaden(grid, flux, f)
        vzs = (-300.0, 0.0, 10.0, 300.0) if tag == "s" else (10.0,)
        out[f"{tag}_vz"] = np.array(vzs)
        for i, vz in enumerate(vzs):
            shifted = T.doppler_shift(grid, vz)
            out[f"{tag}_dop_{i}"] = shifted
            inner = wave[(wave > shifted[3]) & (wave < shifted[-4])]
            out[f"{tag}_resq_{i}"] = inner
            out[f"{tag}_res_{i}"] = T.resample(shifted, flux, inner)
        coeffs = np.array([1.0, 0.01, -0.02, 0.005])
        out[f"{tag}_cheb_c"] = coeffs
        res0 = T.resample(grid, flux, wave)
        out[f"{tag}_cheb_in"] = res0
        out[f"{tag}_cheb"] = T.chebyshev_correct(wave, res0, coeffs)
        out[f"{tag}_renorm"] = np.array(
            [T._get_renorm_factor(wave, res0[0], 1.3 * res0[1] + 0.01)]
        )
    # 1-D input forms and the init-time resample (non log-uniform source grid)
    x = np.sort(rng.uniform(4000, 4100, 257))
    y = np.cos(x / 2.0)
    xq = np.linspace(x[0], x[-1], 101)
    out["irr_x"], out["irr_y"], out["irr_q"] = x, y, xq
    out["irr_out"] = T.resample(x, y, xq)
    np.savez_compressed(os.path.join(OUT, "transforms.npz"), **out)
    print("transforms.npz", len(out))


# ----------------------------------------------------------------------------------- emulator
def gen_emulator():
    out = {}
    for tag, m, custom in (("a", 8, False), ("b", 4, True)):
        o = synth.make_order(N=256, m=m, seed=3)
        var = ls = None
        if custom:  # a "trained-looking" hyper-parameter set
            rng = np.random.default_rng(11)
            var = np.exp(rng.uniform(2, 8, m))
            ls = np.exp(rng.uniform(-0.5, 0.5, (m, 3))) * np.array([300.0, 1.5, 1.5])
        emu, _ = ref_objects(o, variances=var, lengthscales=ls)
        out[f"{tag}_m"] = np.array([m])
        out[f"{tag}_variances"] = emu.variances
        out[f"{tag}_lengthscales"] = emu.lengthscales
        out[f"{tag}_v11"] = emu.v11
        out[f"{tag}_bulk"] = emu.bulk_fluxes
        out[f"{tag}_loglike"] = np.array([emu.log_likelihood()])  # emulator.py:602-619
        # the same after a hyper-parameter update through the trainable-vector interface
        emu2, _ = ref_objects(o, variances=var, lengthscales=ls)
        P = emu2.get_param_vector()
        out[f"{tag}_train_P0"] = P
        out[f"{tag}_train_labels"] = np.array(list(emu2.get_param_dict().keys()))
        emu2.set_param_vector(P + 0.05)
        out[f"{tag}_loglike_shifted"] = np.array([emu2.log_likelihood()])
        queries = np.array(
            [
                [6050.0, 4.2, -0.3],
                [6100.0, 4.5, -0.5],  # exactly a library grid point
                [6000.0, 4.0, -1.0],  # corner of the range
                [6200.0, 5.0, 0.0],
                [6199.9, 4.01, -0.99],
            ]
        )
        out[f"{tag}_queries"] = queries
        for i, q in enumerate(queries):
            mu, cov = emu(q)
            out[f"{tag}_mu_{i}"] = mu
            out[f"{tag}_cov_{i}"] = cov
    np.savez_compressed(os.path.join(OUT, "emulator.npz"), **out)
    print("emulator.npz", len(out))


def gen_emulator_big():
    """Library of the size of the reference's worked example: m = 4, M = 330 (m M = 1320)."""
    o = synth.make_order(N=256, m=4, seed=13, grid_axes=synth.BIG_GRID_AXES)
    emu, _ = ref_objects(o)
    out = {"v11_trace": np.array([np.trace(emu.v11)]), "v11_sum": np.array([emu.v11.sum()]),
           "v11_sample": emu.v11[::97, ::101].copy()}
    queries = np.array([[6050.0, 4.2, -0.3], [6200.0, 4.5, -0.5], [5700.0, 3.5, -2.0], [6699.0, 5.99, -0.01]])
    out["queries"] = queries
    for i, q in enumerate(queries):
        mu, cov = emu(q)
        out[f"mu_{i}"], out[f"cov_{i}"] = mu, cov
    model = ref_model(o)
    ll, logdet, sqmah, flux, cov = parts(model)
    out["lnl"] = np.array([ll, logdet, sqmah])
    out["flux"] = flux
    P = synth.walker_ball(o, B=4, seed=3)
    lls = []
    for p in P:
        model.set_param_vector(p)
        lls.append(model.log_likelihood())
    out["batch_P"], out["batch_lnl"] = P, np.array(lls)
    np.savez_compressed(os.path.join(OUT, "emulator_big.npz"), **out)
    print("emulator_big.npz", ll)


# -------------------------------------------------------------------------------- small models
from gen_golden_cases import COV_ROWS, FULL_COV_CASES, SMALL_CASES, small_case_params  # noqa: E402


def gen_model_small():
    out = {}
    N, m = 256, 4
    o = synth.make_order(N=N, m=m, seed=5)
    factors = 1.0 + 0.01 * np.arange(len(o["grid_points"]))
    out["factors"] = factors
    for name, spec in SMALL_CASES.items():
        c = small_case_params(o, spec)
        model = ref_model(o, params=c, norm=spec.get("norm", False), factors=factors)
        ll, logdet, sqmah, flux, cov = parts(model)
        out[f"{name}_lnl"] = np.array([ll, logdet, sqmah, model._log_scale])
        out[f"{name}_flux"] = flux
        if name in FULL_COV_CASES:
            out[f"{name}_cov"] = cov
        else:  # keep the fixture small: a few full rows + the diagonal
            out[f"{name}_covrows"] = cov[COV_ROWS]
            out[f"{name}_diag"] = cov.diagonal().copy()
        out[f"{name}_labels"] = np.array(model.labels)
        out[f"{name}_vector"] = model.get_param_vector()
    # frozen-cache semantics: freeze global_cov, evaluate, poke the frozen value, evaluate again
    model = ref_model(o, factors=factors)
    model.freeze("global_cov")
    a = model.log_likelihood()
    model["global_cov:log_amp"] = -7.0  # cached matrix must still be used
    b = model.log_likelihood()
    model.thaw("global_cov")
    c_ = model.log_likelihood()
    out["frozen_glob"] = np.array([a, b, c_])
    out["min_dv_wave"] = model.min_dv_wave
    out["bulk_fluxes"] = model.bulk_fluxes
    np.savez_compressed(os.path.join(OUT, "model_small.npz"), **out)
    print("model_small.npz", len(out))


# -------------------------------------------------------------------------------- large models
def sampled(cov, rng, k=1000):
    n = cov.shape[0]
    ii = rng.integers(0, n, k)
    # half of the samples near the diagonal, where the banded kernels live
    jj = np.where(rng.random(k) < 0.5, np.clip(ii + rng.integers(-40, 41, k), 0, n - 1), rng.integers(0, n, k))
    return ii, jj, cov[ii, jj]


def gen_model_large(sizes, nbatch, tag):
    out = {}
    rng = np.random.default_rng(123)
    for N in sizes:
        o = synth.make_order(N=N)
        model = ref_model(o)
        ll, logdet, sqmah, flux, cov = parts(model)
        ii, jj, vals = sampled(cov, rng)
        out[f"n{N}_lnl"] = np.array([ll, logdet, sqmah])
        out[f"n{N}_flux"] = flux
        out[f"n{N}_diag"] = cov.diagonal().copy()
        out[f"n{N}_rowsum"] = cov.sum(axis=1)
        out[f"n{N}_ii"], out[f"n{N}_jj"], out[f"n{N}_vals"] = ii, jj, vals
        del cov
        nb = nbatch.get(N, 0)
        if nb:
            P = synth.walker_ball(o, B=128)[:nb]
            lls = []
            for p in P:
                model.set_param_vector(p)
                lls.append(model.log_likelihood())
            out[f"n{N}_batch_P"] = P
            out[f"n{N}_batch_lnl"] = np.array(lls)
        print("  N", N, ll)
    np.savez_compressed(os.path.join(OUT, f"model_{tag}.npz"), **out)
    print(f"model_{tag}.npz", len(out))


# -------------------------------------------------------------------- cfg 1 (WASP14 plumbing)
def gen_wasp14():
    d = np.load(os.path.join(OUT, "wasp14_order23.npz"))
    mask = d["mask"]
    wave = d["wave"][mask]
    o = synth.make_order(N=len(wave), m=8, seed=2)  # emulator arrays re-built on the WASP14 range
    dv = calculate_dv(wave)
    emu_wl = create_log_lam_grid(dv, 5190.0, 5340.0)["wl"]
    rng = np.random.default_rng(2)
    q, _ = np.linalg.qr(rng.standard_normal((len(emu_wl), 8)))
    o.update(
        emu_wl=emu_wl,
        eigenspectra=np.ascontiguousarray(q.T),
        flux_mean=1 + 0.1 * np.sin(emu_wl / 7),
        flux_std=0.05 + 0.01 * np.cos(emu_wl / 3),
    )
    emu = Emulator(
        o["grid_points"], o["param_names"], o["emu_wl"], o["weights"], o["eigenspectra"],
        o["w_hat"], o["flux_mean"], o["flux_std"], o["factors"],
    )
    emu._trained = True
    data = Spectrum(d["wave"], d["flux"], sigmas=d["sigma"], masks=mask)
    c = synth.centre_params(dict(wave=wave))
    c["vz"] = -4.0
    gp = c.pop("grid_params")
    model = SpectrumModel(emu, data, grid_params=gp, **c)
    ll, logdet, sqmah, flux, cov = parts(model)
    np.savez_compressed(
        os.path.join(OUT, "model_wasp14.npz"),
        emu_wl=emu_wl,
        eigenspectra=o["eigenspectra"],
        flux_mean=o["flux_mean"],
        flux_std=o["flux_std"],
        grid_points=o["grid_points"],
        weights=o["weights"],
        w_hat=o["w_hat"],
        lnl=np.array([ll, logdet, sqmah]),
        flux=flux,
        diag=cov.diagonal().copy(),
        vector=model.get_param_vector(),
        labels=np.array(model.labels),
    )
    print("model_wasp14.npz", ll)


# ------------------------------------------------------------- cfg 3 (25 orders x N = 3000, shared walkers)
def gen_cfg3(n_orders=25, N=3000, nwalk=3):
    """Per-order lnL of the reference SpectrumModel for `nwalk` shared parameter vectors: the multi-order
    likelihood is the sum over orders (docs/intro.rst:71-73; the reference's EchelleModel is a stub)."""
    orders = synth.make_echelle(n_orders, N)
    P = synth.shared_ball(orders[0], B=64)[:nwalk]
    lnl = np.zeros((n_orders, nwalk))
    for o, order in enumerate(orders):
        model = ref_model(order)
        model.freeze("local_cov")  # every order keeps its own local kernel; the walkers share the rest
        assert tuple(model.labels) == synth.SHARED_LABELS, model.labels
        for b, p in enumerate(P):
            model.set_param_vector(p)
            lnl[o, b] = model.log_likelihood()
        print("  order", o, lnl[o])
    np.savez_compressed(
        os.path.join(OUT, "model_cfg3.npz"), P=P, lnl=lnl, labels=np.array(synth.SHARED_LABELS),
        n_orders=np.array([n_orders]), N=np.array([N]), seed0=np.array([100]),
    )
    print("model_cfg3.npz", lnl.sum(axis=0))


# ------------------------------------- full BASELINE batches: the LAST walker of each config's bench batch
def gen_fullbatch():
    """The parity tests at the BASELINE batch sizes (cfg 2: 128 walkers, cfg 3: 25 orders x 64, cfg 5: 32) compare the
    first walkers with model_cfg2 / model_cfg3 and the LAST walker of each batch (plus walker 0 of cfg 5's ball)
    with these values of the reference."""
    import time

    out = {}
    o = synth.make_order(N=4096)
    model = ref_model(o)
    P = synth.walker_ball(o, B=128)
    model.set_param_vector(P[127])
    out["cfg2_P127"], out["cfg2_lnl127"] = P[127], np.array([model.log_likelihood()])
    print("  cfg2 walker 127", out["cfg2_lnl127"])
    orders = synth.make_echelle(25, 3000)
    Ps = synth.shared_ball(orders[0], B=64)
    lnl = np.zeros(25)
    for k, order in enumerate(orders):
        model = ref_model(order)
        model.freeze("local_cov")
        model.set_param_vector(Ps[63])
        lnl[k] = model.log_likelihood()
    out["cfg3_P63"], out["cfg3_lnl63"] = Ps[63], lnl
    print("  cfg3 walker 63", lnl.sum())
    o = synth.make_order(N=16384)
    model = ref_model(o)
    P = synth.walker_ball(o, B=32)
    vals = []
    for b in (0, 31):
        t0 = time.time()
        model.set_param_vector(P[b])
        vals.append(model.log_likelihood())
        print("  cfg5 walker", b, vals[-1], f"{time.time() - t0:.0f} s")
    out["cfg5_P"], out["cfg5_lnl"] = P[[0, 31]], np.array(vals)
    np.savez_compressed(os.path.join(OUT, "model_fullbatch.npz"), **out)
    print("model_fullbatch.npz", len(out))


def gen_emulator_train_big():
    """Emulator.log_likelihood() (emulator.py:602-619) at the worked-example size m = 4, M = 330 (a 1320 x 1320
    cho_factor per objective call, examples/setup.ipynb:47,185,215) for two hyper-parameter vectors."""
    o = synth.make_order(N=256, m=4, seed=13, grid_axes=synth.BIG_GRID_AXES)
    emu, _ = ref_objects(o)
    P0 = emu.get_param_vector()
    out = {"P0": P0, "labels": np.array(list(emu.get_param_dict().keys())), "lnl0": np.array([emu.log_likelihood()])}
    rng = np.random.default_rng(17)
    P1 = P0 + rng.uniform(-0.3, 0.3, len(P0))
    emu.set_param_vector(P1)
    out["P1"], out["lnl1"] = P1, np.array([emu.log_likelihood()])
    out["v11_trace1"] = np.array([np.trace(emu.v11)])
    np.savez_compressed(os.path.join(OUT, "emulator_train_big.npz"), **out)
    print("emulator_train_big.npz", out["lnl0"], out["lnl1"])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    big = "--big" in sys.argv
    only = [a for a in sys.argv[1:] if not a.startswith("--")]

    def want(name):
        return not only or name in only

    if want("kernels"):
        gen_kernels()
    if want("transforms"):
        gen_transforms()
    if want("emulator"):
        gen_emulator()
    if want("small"):
        gen_model_small()
    if want("wasp14"):
        gen_wasp14()
    if want("large"):
        gen_model_large([1024, 3000], {1024: 8, 3000: 2}, "large")
    if "cfg3" in only:
        gen_cfg3()
    if "emulator_big" in only:
        gen_emulator_big()
    if "emulator_train_big" in only:
        gen_emulator_train_big()
    if "fullbatch" in only:
        gen_fullbatch()
    if big:
        gen_model_large([4096], {4096: 8}, "cfg2")
        gen_model_large([16384], {}, "cfg5")
