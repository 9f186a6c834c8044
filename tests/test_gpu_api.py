"""The drop-in Python surface (SpectrumModel / Emulator / EchelleModel) on the GPU against values
produced by the real reference (golden fixtures) and against the oracle.  Run with -m gpu."""
import os
import sys

import numpy as np
import pytest

from conftest import load_golden
from starfish_amd import Spectrum, synth
from starfish_amd.emulator import Emulator
from starfish_amd.models import EchelleModel, SpectrumModel

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import gen_golden_cases as G  # noqa: E402

pytestmark = pytest.mark.gpu


def build(o, params=None, norm=False, factors=None):
    emu = Emulator(o["grid_points"], o["param_names"], o["emu_wl"], o["weights"], o["eigenspectra"],
                   o["w_hat"], o["flux_mean"], o["flux_std"], o["factors"] if factors is None else factors)
    emu._trained = True
    data = Spectrum(o["wave"], o["flux"], sigmas=o["sigma"])
    c = dict(synth.centre_params(o)) if params is None else dict(params)
    gp = c.pop("grid_params")
    return SpectrumModel(emu, data, grid_params=gp, norm=norm, **c)


def close(a, b):
    return abs(a - b) <= 1e-8 * abs(b) + 1e-8


@pytest.mark.parametrize("name", list(G.SMALL_CASES))
def test_spectrum_model_cases_vs_reference(name):
    g = load_golden("model_small.npz")
    o = synth.make_order(N=256, m=4, seed=5)
    spec = G.SMALL_CASES[name]
    m = build(o, G.small_case_params(o, spec), norm=spec.get("norm", False), factors=g["factors"])
    assert tuple(g[f"{name}_labels"]) == m.labels
    np.testing.assert_allclose(m.get_param_vector(), g[f"{name}_vector"])
    np.testing.assert_allclose(m.min_dv_wave, g["min_dv_wave"], rtol=1e-15)
    np.testing.assert_allclose(m.bulk_fluxes, g["bulk_fluxes"], rtol=0, atol=1e-12)
    ref = g[f"{name}_lnl"]
    assert close(m.log_likelihood(), ref[0])
    assert abs(m._log_scale - ref[3]) <= 1e-10 * max(1, abs(ref[3]))
    assert len(m.residuals) == 1 and m._lnprob is not None
    flux, cov = m()
    np.testing.assert_allclose(flux, g[f"{name}_flux"], rtol=0, atol=1e-10)
    assert cov.shape == (256, 256)


def test_frozen_covariance_cache_semantics():
    g = load_golden("model_small.npz")
    o = synth.make_order(N=256, m=4, seed=5)
    m = build(o, factors=g["factors"])
    m.freeze("global_cov")
    assert m._glob_cov is None
    a = m.log_likelihood()
    assert m._glob_cov is not None
    m["global_cov:log_amp"] = -7.0  # the cached (frozen) kernel must still be used
    b = m.log_likelihood()
    m.thaw("global_cov")
    c = m.log_likelihood()
    want = g["frozen_glob"]
    assert close(a, want[0]) and close(b, want[1]) and close(c, want[2])
    assert a == b and c != a


def test_emcee_style_loop_and_batch_agree():
    """The reference's driver: log_prob(P) = set_param_vector(P); log_likelihood(priors)
    (examples/single.ipynb:458-460) -- and the batched front-end gives the same numbers."""
    import scipy.stats as st

    g = load_golden("model_large.npz")
    o = synth.make_order(N=1024)
    m = build(o)
    priors = {"vsini": st.uniform(0, 500), "T": st.norm(6050, 100), "cheb:1": st.uniform(-3, 6)}
    P = g["n1024_batch_P"][:6]

    def log_prob(p):
        m.set_param_vector(p)
        return m.log_likelihood(priors)

    serial = np.array([log_prob(p) for p in P])
    prior_terms = np.array([sum(pr.logpdf(p[m.labels.index(k)]) for k, pr in priors.items()) for p in P])
    for b in range(6):
        assert close(serial[b] - prior_terms[b], g["n1024_batch_lnl"][b])
    batch = m.log_likelihood_batch(P, priors)
    np.testing.assert_allclose(batch, serial, rtol=1e-12)
    # non-finite prior -> -inf without touching the device; out-of-grid walker -> -inf + info
    bad = P.copy()
    bad[1, m.labels.index("vsini")] = -5.0
    bad[2, m.labels.index("T")] = 7000.0
    ll, info = m.log_likelihood_batch(bad, priors, return_info=True)
    assert ll[1] == -np.inf and ll[2] == -np.inf and info[2] == -1 and np.isfinite(ll[0])


def test_structured_solver_option_matches_the_dense_model():
    """solver="auto" (band + rank-m Woodbury, dense fallback) is a drop-in for the default dense solve:
    same values from the reference's driver loop, the batched front-end, frozen caches and the sampler."""
    import scipy.stats as st

    g = load_golden("model_large.npz")
    o = synth.make_order(N=1024)
    dense, auto = build(o), build(o)
    auto.solver = "auto"
    priors = {"vsini": st.uniform(0, 500), "T": st.norm(6050, 100)}
    P = g["n1024_batch_P"][:6]
    a = auto.log_likelihood_batch(P, priors)
    np.testing.assert_allclose(a, dense.log_likelihood_batch(P, priors), rtol=1e-11)
    for b in range(6):
        prior = sum(pr.logpdf(P[b][auto.labels.index(k)]) for k, pr in priors.items())
        assert close(a[b] - prior, g["n1024_batch_lnl"][b])
    auto.set_param_vector(P[0])
    assert close(auto.log_likelihood(), g["n1024_batch_lnl"][0])
    np.testing.assert_allclose(auto.residuals[-1], dense_resid(dense, P[0]), rtol=0, atol=1e-12)
    # frozen covariance hyper-parameters: the cached values are the ones the band is built from
    auto.freeze("global_cov")
    dense.set_param_vector(P[0])
    dense.freeze("global_cov")
    v0, d0 = auto.log_likelihood(), dense.log_likelihood()
    auto["global_cov:log_amp"] = -7.0
    dense["global_cov:log_amp"] = -7.0
    assert auto.log_likelihood() == v0 and dense.log_likelihood() == d0
    assert close(v0, d0)
    # a length scale too wide for the device window silently takes the dense route
    auto.thaw("global_cov")
    dense.thaw("global_cov")
    auto["global_cov:log_ls"] = dense["global_cov:log_ls"] = float(np.log(300.0))
    assert close(auto.log_likelihood(), dense.log_likelihood())
    with pytest.raises(ValueError):
        build(o).__class__(auto.emulator, Spectrum(o["wave"], o["flux"], sigmas=o["sigma"]), [6050.0, 4.2, -0.3],
                           solver="fast")


def dense_resid(model, p):
    model.set_param_vector(p)
    model.log_likelihood()
    return model.residuals[-1]


def test_scalar_path_raises_like_the_reference():
    o = synth.make_order(N=256, m=4, seed=5)
    m = build(o)
    m["T"] = 7000.0
    with pytest.raises(ValueError):
        m.log_likelihood()
    m["T"] = 6050.0
    m["vsini"] = -1.0
    with pytest.raises(ValueError):
        m.log_likelihood()
    with pytest.raises(ValueError):
        m()


def test_emulator_call_api():
    g = load_golden("emulator.npz")
    o = synth.make_order(N=256, m=8, seed=3)
    emu = Emulator(o["grid_points"], o["param_names"], o["emu_wl"], o["weights"], o["eigenspectra"],
                   o["w_hat"], o["flux_mean"], o["flux_std"], o["factors"])
    np.testing.assert_allclose(emu.v11, g["a_v11"], rtol=1e-12, atol=1e-12)
    with pytest.warns(UserWarning):
        mu, cov = emu(g["a_queries"][0])
    emu._trained = True
    np.testing.assert_allclose(mu, g["a_mu_0"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(cov, g["a_cov_0"], rtol=1e-9, atol=1e-9)
    mu2, var2 = emu(g["a_queries"][:3], full_cov=False, reinterpret_batch=True)
    assert mu2.shape == (3, 8) and var2.shape == (3, 8)
    np.testing.assert_allclose(var2[1], np.diag(g["a_cov_1"]), rtol=1e-9)
    with pytest.raises(ValueError):
        emu([5000.0, 4.2, -0.3])
    with pytest.raises(ValueError):
        emu(g["a_queries"][:2], full_cov=True, reinterpret_batch=True)
    np.testing.assert_allclose(emu.bulk_fluxes, g["a_bulk"])


def test_echelle_model_sums_orders():
    orders = [synth.make_order(N=256, m=4, seed=5, wave0=5000.0 * 1.02**k) for k in range(3)]
    o0 = orders[0]
    emu_wl = np.concatenate([o["emu_wl"] for o in orders])
    # one emulator spanning all orders: re-use order 0's recipe on a wide grid
    wide = synth.make_order(N=256, m=4, seed=5, wave0=5000.0, pad=20.0)
    from starfish_amd.utils import create_log_lam_grid

    wl = create_log_lam_grid(2.0, emu_wl.min(), emu_wl.max())["wl"]
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.standard_normal((len(wl), 4)))
    emu = Emulator(wide["grid_points"], wide["param_names"], wl, wide["weights"], np.ascontiguousarray(q.T),
                   wide["w_hat"], 1 + 0.1 * np.sin(wl / 7), 0.05 + 0.01 * np.cos(wl / 3), wide["factors"])
    emu._trained = True
    waves = np.vstack([o["wave"] for o in orders])
    fluxes = np.vstack([o["flux"] for o in orders])
    sig = np.vstack([o["sigma"] for o in orders])
    data = Spectrum(waves, fluxes, sig)
    shared = dict(vz=10.0, vsini=30.0, log_scale=0.0, global_cov=dict(log_amp=-9.0, log_ls=np.log(10.0)),
                  cheb=[0.01, -0.02])
    em = EchelleModel(emu, data, [6050.0, 4.2, -0.3], **shared)
    total = em.log_likelihood()
    parts = [SpectrumModel(emu, Spectrum(waves[k], fluxes[k], sig[k]), [6050.0, 4.2, -0.3],
                           **{k2: (dict(v) if isinstance(v, dict) else v) for k2, v in shared.items()}).log_likelihood()
             for k in range(3)]
    assert close(total, sum(parts))
    P = np.tile(em.get_param_vector(), (4, 1))
    P[:, em.labels.index("vz")] += [0.0, 0.1, -0.1, 0.2]
    batch = em.log_likelihood_batch(P)
    assert close(batch[0], total) and batch.shape == (4,)


@pytest.mark.parametrize("tag,m", [("a", 8), ("b", 4)])
def test_emulator_training_likelihood_vs_reference(tag, m):
    """SURVEY f-4: Emulator.log_likelihood on the same Cholesky kernels, against the reference's values."""
    g = load_golden("emulator.npz")
    o = synth.make_order(N=256, m=m, seed=3)
    emu = Emulator(o["grid_points"], o["param_names"], o["emu_wl"], o["weights"], o["eigenspectra"],
                   o["w_hat"], o["flux_mean"], o["flux_std"], o["factors"],
                   variances=g[f"{tag}_variances"], lengthscales=g[f"{tag}_lengthscales"])
    assert list(g[f"{tag}_train_labels"]) == list(emu.get_param_dict().keys())
    np.testing.assert_allclose(emu.get_param_vector(), g[f"{tag}_train_P0"], rtol=1e-14)
    want = g[f"{tag}_loglike"][0]
    assert abs(emu.log_likelihood() - want) <= 1e-9 * abs(want)
    emu.set_param_vector(g[f"{tag}_train_P0"] + 0.05)
    want = g[f"{tag}_loglike_shifted"][0]
    assert abs(emu.log_likelihood() - want) <= 1e-9 * abs(want)
    with pytest.raises(ValueError):
        emu.set_param_vector(g[f"{tag}_train_P0"][:-1])


def test_emulator_train_improves_likelihood():
    """Mirrors the reference's tests/test_emulator/test_emulator.py:87-98 on a short optimisation."""
    o = synth.make_order(N=256, m=2, seed=3)
    emu = Emulator(o["grid_points"], o["param_names"], o["emu_wl"], o["weights"], o["eigenspectra"],
                   o["w_hat"], o["flux_mean"], o["flux_std"], o["factors"])
    before = emu.log_likelihood()
    emu.train(options=dict(maxiter=40))
    assert emu.log_likelihood() >= before


def test_realistic_emulator_size_vs_reference():
    """a library of the reference's worked-example size (m = 4, M = 330 -> m M = 1320,
    examples/setup.ipynb:47,185): Emulator.__call__ and the model likelihood against the reference's values."""
    g = load_golden("emulator_big.npz")
    o = synth.make_order(N=256, m=4, seed=13, grid_axes=synth.BIG_GRID_AXES)
    assert len(o["grid_points"]) == 330
    m = build(o)
    emu = m.emulator
    np.testing.assert_allclose(np.trace(emu.v11), g["v11_trace"][0], rtol=1e-13)
    np.testing.assert_allclose(emu.v11[::97, ::101], g["v11_sample"], rtol=1e-12, atol=1e-300)
    for i, q in enumerate(g["queries"]):
        mu, cov = emu(q)
        np.testing.assert_allclose(mu, g[f"mu_{i}"], rtol=1e-9, atol=1e-9 * np.abs(g[f"mu_{i}"]).max())
        np.testing.assert_allclose(cov, g[f"cov_{i}"], rtol=1e-9, atol=1e-9 * np.abs(g[f"cov_{i}"]).max())
    assert close(m.log_likelihood(), g["lnl"][0])
    got = m.log_likelihood_batch(g["batch_P"])
    assert all(close(a, b) for a, b in zip(got, g["batch_lnl"]))
    # the library's own (scalar host) factorisation of v11 gives the same constants as the LAPACK factor handed in
    from starfish_amd import _device as D

    dev = m._device()
    plain = D.DeviceOrder(*dev._keep[:10])
    md, rows = m._pack(g["batch_P"], update_caches=False)[1:]
    np.testing.assert_allclose(plain.loglike(md, rows)["lnl"], dev.loglike(md, rows)["lnl"], rtol=1e-12)


def test_emulator_training_likelihood_worked_example_size_vs_reference():
    """Emulator.log_likelihood() at the size of the reference's worked example (m = 4, M = 330: one
    1320 x 1320 Cholesky per objective call of Emulator.train, emulator.py:484-524,602-619) against the reference's
    values for two hyper-parameter vectors (tests/golden/emulator_train_big.npz)."""
    g = load_golden("emulator_train_big.npz")
    o = synth.make_order(N=256, m=4, seed=13, grid_axes=synth.BIG_GRID_AXES)
    emu = Emulator(o["grid_points"], o["param_names"], o["emu_wl"], o["weights"], o["eigenspectra"],
                   o["w_hat"], o["flux_mean"], o["flux_std"], o["factors"])
    assert emu.v11.shape == (1320, 1320)
    assert list(g["labels"]) == list(emu.get_param_dict().keys())
    np.testing.assert_allclose(emu.get_param_vector(), g["P0"], rtol=1e-14)
    want = g["lnl0"][0]
    assert abs(emu.log_likelihood() - want) <= 1e-9 * abs(want)
    emu.set_param_vector(g["P1"])
    np.testing.assert_allclose(np.trace(emu.v11), g["v11_trace1"][0], rtol=1e-13)
    want = g["lnl1"][0]
    got = emu.log_likelihood()
    assert abs(got - want) <= 1e-9 * abs(want), (got, want)
    assert got == emu.log_likelihood()  # deterministic


def test_device_built_v11_matches_the_host_matrix():
    """sf_emulator_v11_build (the training objective's matrix, built from the hyper-parameters on the device) against the
    host numpy build the queries use (emulator.py:126-128; kernels.py:5-49), identity padding included; a matrix assigned
    by hand is factored as it is; changing a hyper-parameter drops it."""
    import torch

    from starfish_amd import _device as D
    from starfish_amd import _lib

    o = synth.make_order(N=256, m=4, seed=3)
    rng = np.random.default_rng(11)
    emu = Emulator(o["grid_points"], o["param_names"], o["emu_wl"], o["weights"], o["eigenspectra"], o["w_hat"],
                   o["flux_mean"], o["flux_std"], o["factors"], variances=np.exp(rng.uniform(2, 8, 4)),
                   lengthscales=np.exp(rng.uniform(-0.5, 0.5, (4, 3))) * np.array([300.0, 1.5, 1.5]), lambda_xi=1.7)
    lib = _lib.require_gpu()
    dev = D.device_of()
    M, P = emu.grid_points.shape
    n = 4 * M
    npad = -(-n // 64) * 64
    lda = npad + 16
    A = torch.full((npad, lda), np.nan, dtype=torch.float64, device=dev)
    hyper = D.to_dev(np.concatenate([[emu.lambda_xi], emu.variances, emu.lengthscales.ravel()]), dev)
    d_grid, d_iphiphi = D.to_dev(emu.grid_points, dev), D.to_dev(emu.iPhiPhi, dev)  # (kept alive across the launch)
    _lib.check(lib.sf_emulator_v11_build(D.ptr(d_grid), M, P, 4, D.ptr(hyper), D.ptr(d_iphiphi), D.ptr(A), npad, lda,
                                         D.stream_ptr(dev)))
    got = A.cpu().numpy()
    np.testing.assert_allclose(got[:n, :n], emu.v11, rtol=1e-13, atol=1e-300)
    np.testing.assert_array_equal(got[n:, :npad], np.eye(npad)[n:])
    assert not got[:n, n:npad].any() and np.isnan(got[:, npad:]).all()  # the 16 spare columns of a row are never touched
    base = emu.log_likelihood()
    emu.v11 = emu.v11 + 0.5 * np.eye(n)  # assigned by hand: factored as given
    shifted = emu.log_likelihood()
    from scipy.linalg import cho_factor, cho_solve

    f = cho_factor(emu.v11)
    want = -(2 * np.sum(np.log(f[0].diagonal())) + emu.w_hat @ cho_solve(f, emu.w_hat)) / 2
    assert abs(shifted - want) <= 1e-10 * abs(want) and shifted != base
    emu.set_param_vector(emu.get_param_vector())  # hyper-parameters rule again
    assert abs(emu.log_likelihood() - base) <= 1e-12 * abs(base)
