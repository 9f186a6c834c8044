"""Parity of the structure-exploiting solver (band + rank-m Woodbury) against
numpy, the CPU oracle, the reference-generated goldens and the dense HIP path.  Needs an MI355X."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import load_golden
from oracle import sf_oracle as O
from starfish_amd import _device as D
from starfish_amd import _lib, synth

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import gen_golden_cases as G  # noqa: E402
from gpu_helpers import device_order, oracle_order, pack_rows  # noqa: E402
from test_gpu_model import close_lnl, small_params  # noqa: E402

pytestmark = pytest.mark.gpu


def random_band_spd(rng, n, w):
    """Dense symmetric positive definite matrix with half-bandwidth w, plus its lower band storage."""
    A = np.zeros((n, n))
    for d in range(1, w + 1):
        v = rng.standard_normal(n - d) * 0.3
        A[np.arange(d, n), np.arange(n - d)] = v
    A = A + A.T
    A[np.diag_indices(n)] = np.abs(A).sum(axis=1) + 0.5 + rng.random(n)
    ldb = w + 1 + (w + 1) % 2
    band = np.zeros((n, ldb))
    for d in range(w + 1):
        band[d:, d] = A[np.arange(d, n), np.arange(n - d)]
    return A, band, ldb


def run_band_forms(bands, ldb, n, w, rhs):
    import torch

    lib = _lib.require_gpu()
    dev = D.device_of()
    batch, nrhs = rhs.shape[0], rhs.shape[1]
    d_band = D.to_dev(bands, dev)
    d_rhs = D.to_dev(rhs, dev)
    logdet = D.empty((batch,), dev)
    gram = D.empty((batch, nrhs, nrhs), dev)
    info = D.empty((batch,), dev, torch.int32)
    rc = lib.sf_band_logdet_gram_batch(
        D.ptr(d_band), n, w, ldb, n * ldb, batch, D.ptr(d_rhs), nrhs, n, nrhs * n, D.ptr(logdet), D.ptr(gram),
        D.ptr(info), D.stream_ptr(dev),
    )
    _lib.check(rc, "sf_band_logdet_gram_batch")
    return logdet.cpu().numpy(), gram.cpu().numpy(), info.cpu().numpy()


@pytest.mark.parametrize("n,w,nrhs", [(16, 0, 1), (100, 5, 9), (257, 16, 9), (257, 37, 20), (1000, 64, 9),
                                       (333, 80, 33), (64, 63, 3), (40, 60, 2), (2048, 31, 9), (500, 144, 9), (300, 128, 17)])
def test_band_logdet_gram_vs_numpy(n, w, nrhs):
    rng = np.random.default_rng(100 * n + w)
    batch = 3
    mats, bands = [], []
    w_eff = min(w, n - 1)
    for _ in range(batch):
        A, band, ldb = random_band_spd(rng, n, w_eff)
        if w_eff < w:  # storage wider than the matrix: extra diagonals are zero
            ldb2 = w + 1 + (w + 1) % 2
            b2 = np.zeros((n, ldb2))
            b2[:, : band.shape[1]] = band
            band, ldb = b2, ldb2
        mats.append(A)
        bands.append(band)
    rhs = rng.standard_normal((batch, nrhs, n))
    logdet, gram, info = run_band_forms(np.stack(bands), ldb, n, w, rhs)
    assert (info == 0).all()
    for b in range(batch):
        sign, want_ld = np.linalg.slogdet(mats[b])
        assert sign > 0
        assert abs(logdet[b] - want_ld) <= 1e-12 * max(1.0, abs(want_ld))
        want = rhs[b] @ np.linalg.solve(mats[b], rhs[b].T)
        np.testing.assert_allclose(gram[b], want, rtol=1e-11, atol=1e-12 * np.abs(want).max())
        np.testing.assert_array_equal(gram[b], gram[b].T)


def test_band_not_positive_definite_reports_first_bad_pivot():
    rng = np.random.default_rng(3)
    n, w = 200, 7
    A, band, ldb = random_band_spd(rng, n, w)
    band[77, 0] = -1.0  # pivot 78 (1-based) becomes negative
    rhs = rng.standard_normal((1, 2, n))
    _, _, info = run_band_forms(band[None], ldb, n, w, rhs)
    assert info[0] == 78


@pytest.fixture(scope="module")
def small():
    o = synth.make_order(N=256, m=4, seed=5)
    oo = oracle_order(o)
    return o, oo, device_order(oo)


@pytest.mark.parametrize("name", list(G.SMALL_CASES))
def test_small_model_cases_banded_vs_reference(small, name):
    o, oo, do = small
    g = load_golden("model_small.npz")
    p = small_params(o, name, g["factors"])
    md, rows = pack_rows(do, [p])
    ref = g[f"{name}_lnl"]
    hw = do.halfwidth_bound(md, rows)
    # the host bound really bounds the support of the structured part of the reference covariance
    f_or, c_or, _ = O.forward_model(oo, p)
    X = None
    if "global_cov" in p or "local_cov" in p:
        K = np.zeros_like(c_or)
        if "global_cov" in p:
            K += O.matern32_global(oo.wave, np.exp(p["global_cov"][0]), np.exp(p["global_cov"][1]))
        for mu, la, ls in p.get("local_cov", []):
            K += O.gaussian_local(oo.wave, np.exp(la), mu, np.exp(ls))
        ii, jj = np.nonzero(K)
        true_hw = int(np.abs(ii - jj).max()) if ii.size else 0
        assert true_hw <= hw[0]
        X = true_hw
    if hw[0] > do.banded_max_halfwidth():
        out = do.loglike(md, rows, solver="banded")
        assert out["info"][0] == D.INFO_BANDWIDTH and out["lnl"][0] == -np.inf
        out = do.loglike(md, rows, solver="auto")  # falls back to the dense factorisation
    else:
        out = do.loglike(md, rows, solver="banded", want_resid=True)
        np.testing.assert_allclose(out["resid"][0], g[f"{name}_flux"] - oo.flux, rtol=0, atol=1e-10)
    assert out["info"][0] == 0, (hw, X)
    assert close_lnl(out["lnl"][0], ref[0])
    assert abs(out["logdet"][0] - ref[1]) <= 1e-10 * abs(ref[1])
    assert abs(out["sqmah"][0] - ref[2]) <= 1e-8 * abs(ref[2])
    assert abs(out["log_scale"][0] - ref[3]) <= 1e-10 * max(1.0, abs(ref[3]))


def test_banded_batch_matches_reference_and_dense_n1024():
    g = load_golden("model_large.npz")
    o = synth.make_order(N=1024)
    oo = oracle_order(o)
    do = device_order(oo)
    P = synth.walker_ball(o, B=128)
    plist = [synth.vector_to_oracle_params(p) for p in P[:32]]
    md, rows = pack_rows(do, plist)
    dense = do.loglike(md, rows)
    band = do.loglike(md, rows, solver="banded")
    assert (band["info"] == 0).all()
    for b in range(8):
        assert close_lnl(band["lnl"][b], g["n1024_batch_lnl"][b])
    np.testing.assert_allclose(band["lnl"], dense["lnl"], rtol=1e-11)
    np.testing.assert_allclose(band["logdet"], dense["logdet"], rtol=1e-12)
    np.testing.assert_allclose(band["sqmah"], dense["sqmah"], rtol=1e-9)
    # chunking and batch position do not matter
    b2 = do.loglike(md, rows[::-1].copy(), solver="banded", max_chunk=7)
    np.testing.assert_array_equal(b2["lnl"][::-1], band["lnl"])


def test_banded_halfwidth_too_small_is_flagged_and_auto_recovers():
    o = synth.make_order(N=1024)
    oo = oracle_order(o)
    do = device_order(oo)
    import torch

    plist = [synth.vector_to_oracle_params(p) for p in synth.walker_ball(o, B=4)]
    md, rows = pack_rows(do, plist)
    hw = int(do.halfwidth_bound(md, rows).max())
    P = D.to_dev(rows, do.dev)
    lnl = D.empty((4,), do.dev)
    info = D.empty((4,), do.dev, torch.int32)
    do.loglike_banded_device(md, P, 16, lnl, info)  # far below the true support (~60 px)
    assert (info.cpu().numpy() == D.INFO_BANDWIDTH).all() and (lnl.cpu().numpy() == -np.inf).all()
    do.loglike_banded_device(md, P, hw, lnl, info)
    assert (info.cpu().numpy() == 0).all()
    want = do.loglike(md, rows)["lnl"]
    np.testing.assert_allclose(lnl.cpu().numpy(), want, rtol=1e-11)
    # a walker with a huge length scale cannot use the window: "auto" sends only that one to the dense path
    wide = dict(plist[0], global_cov=(plist[0]["global_cov"][0], np.log(400.0)))
    md, rows = pack_rows(do, [plist[0], wide, plist[1]])
    assert do.halfwidth_bound(md, rows)[1] > do.banded_max_halfwidth()
    auto = do.loglike(md, rows, solver="auto")
    dense = do.loglike(md, rows)
    assert (auto["info"] == 0).all()
    np.testing.assert_allclose(auto["lnl"], dense["lnl"], rtol=1e-11)
    only = do.loglike(md, rows, solver="banded")
    assert only["info"].tolist() == [0, D.INFO_BANDWIDTH, 0]


def test_banded_cfg2_full_size_vs_reference():
    g = load_golden("model_cfg2.npz")
    o = synth.make_order(N=4096)
    oo = oracle_order(o)
    do = device_order(oo)
    P = g["n4096_batch_P"]
    plist = [synth.vector_to_oracle_params(synth.centre_vector(o))] + [synth.vector_to_oracle_params(p) for p in P]
    md, rows = pack_rows(do, plist)
    out = do.loglike(md, rows, solver="banded")
    assert (out["info"] == 0).all()
    assert close_lnl(out["lnl"][0], g["n4096_lnl"][0])
    assert abs(out["lnl"][0] - 16579.1706341206) < 1e-6
    assert abs(out["logdet"][0] - g["n4096_lnl"][1]) <= 1e-10 * abs(g["n4096_lnl"][1])
    assert abs(out["sqmah"][0] - g["n4096_lnl"][2]) <= 1e-8 * abs(g["n4096_lnl"][2])
    for b in range(len(P)):
        assert close_lnl(out["lnl"][1 + b], g["n4096_batch_lnl"][b])


def test_banded_cfg5_and_wasp14_vs_reference():
    g = load_golden("model_cfg5.npz")
    o = synth.make_order(N=16384)
    oo = oracle_order(o)
    do = device_order(oo)
    p = synth.vector_to_oracle_params(synth.centre_vector(o))
    md, rows = pack_rows(do, [p])
    out = do.loglike(md, rows, solver="banded")
    assert out["info"][0] == 0
    assert close_lnl(out["lnl"][0], g["n16384_lnl"][0])
    assert abs(out["logdet"][0] - g["n16384_lnl"][1]) <= 1e-10 * abs(g["n16384_lnl"][1])
    del do

    d = load_golden("wasp14_order23.npz")
    g = load_golden("model_wasp14.npz")
    mask = d["mask"]
    oo = O.OracleOrder(d["wave"][mask], d["flux"][mask], d["sigma"][mask], g["emu_wl"], g["eigenspectra"],
                       g["flux_mean"], g["flux_std"], g["grid_points"], g["w_hat"])
    do = device_order(oo)
    p = synth.vector_to_oracle_params(g["vector"])
    md, rows = pack_rows(do, [p])
    out = do.loglike(md, rows, solver="auto")
    assert out["info"][0] == 0
    assert close_lnl(out["lnl"][0], g["lnl"][0])


@pytest.mark.parametrize("ls", [14.0, 25.0, 60.0])
def test_wide_band_kernel_matches_dense_and_oracle(ls):
    """Length scales beyond the LDS window (W = 24 ls/dv > 144 px) go through the in-place left-looking
    band kernel: same value as the dense path and the oracle."""
    o = synth.make_order(N=1024)
    oo = oracle_order(o)
    do = device_order(oo)
    plist = [synth.vector_to_oracle_params(p) for p in synth.walker_ball(o, B=6)]
    for p in plist:
        p["global_cov"] = (p["global_cov"][0], float(np.log(ls)) + 0.01 * (p["vz"] - 10.0))
    md, rows = pack_rows(do, plist)
    hw = do.halfwidth_bound(md, rows)
    assert (hw > do.banded_window_halfwidth()).all() and (hw <= do.banded_max_halfwidth()).all()
    band = do.loglike(md, rows, solver="banded", want_resid=True)
    dense = do.loglike(md, rows, want_resid=True)
    assert (band["info"] == 0).all()
    np.testing.assert_allclose(band["lnl"], dense["lnl"], rtol=1e-10)
    np.testing.assert_allclose(band["logdet"], dense["logdet"], rtol=1e-11)
    np.testing.assert_allclose(band["sqmah"], dense["sqmah"], rtol=1e-8)
    np.testing.assert_array_equal(band["resid"], dense["resid"])
    for b in (0, 5):
        assert close_lnl(band["lnl"][b], O.log_likelihood(oo, plist[b]))
    # mixed widths in one call: narrow walkers use the window sweep, wide ones this kernel, absurdly wide -> dense
    mixed = [dict(plist[0], global_cov=(plist[0]["global_cov"][0], np.log(5.0))), plist[1],
             dict(plist[2], global_cov=(plist[2]["global_cov"][0], np.log(900.0)))]
    md, rows = pack_rows(do, mixed)
    auto = do.loglike(md, rows, solver="auto")
    ref = do.loglike(md, rows)
    assert (auto["info"] == 0).all()
    np.testing.assert_allclose(auto["lnl"], ref["lnl"], rtol=1e-10)


def test_wide_band_full_size_cfg2_shape():
    o = synth.make_order(N=4096)
    oo = oracle_order(o)
    do = device_order(oo)
    plist = [synth.vector_to_oracle_params(p) for p in synth.walker_ball(o, B=4)]
    for p in plist:
        p["global_cov"] = (p["global_cov"][0], float(np.log(30.0)))
    md, rows = pack_rows(do, plist)
    band = do.loglike(md, rows, solver="banded")
    dense = do.loglike(md, rows)
    assert (band["info"] == 0).all()
    np.testing.assert_allclose(band["lnl"], dense["lnl"], rtol=1e-10)


@pytest.mark.parametrize("N", [1000, 3000])
def test_auto_solver_fuzz_ragged_sizes(N):
    """Random hyper-parameters (supports from a few pixels to wider than any banded kernel), sizes that are not
    multiples of 16: solver="auto" agrees with the dense factorisation walker by walker."""
    rng = np.random.default_rng(N)
    o = synth.make_order(N=N)
    oo = oracle_order(o)
    do = device_order(oo)
    plist = [synth.vector_to_oracle_params(p) for p in synth.walker_ball(o, B=24, seed=3)]
    for i, p in enumerate(plist):
        ls = float(np.exp(rng.uniform(np.log(0.3), np.log(70.0))))
        sig = float(np.exp(rng.uniform(np.log(1.0), np.log(40.0))))
        mu = float(rng.uniform(oo.wave[0] - 1.0, oo.wave[-1] + 1.0))  # patches may hang over either edge
        p["global_cov"] = (p["global_cov"][0] + rng.uniform(-1, 1), np.log(ls))
        p["local_cov"] = [(mu, p["local_cov"][0][1] + rng.uniform(-1, 1), np.log(sig))]
    md, rows = pack_rows(do, plist)
    hw = do.halfwidth_bound(md, rows)
    assert (hw <= do.banded_window_halfwidth()).any() and (hw > do.banded_window_halfwidth()).any()
    auto = do.loglike(md, rows, solver="auto")
    dense = do.loglike(md, rows)
    assert (auto["info"] == 0).all() and (dense["info"] == 0).all()
    np.testing.assert_allclose(auto["lnl"], dense["lnl"], rtol=1e-10)
    np.testing.assert_allclose(auto["logdet"], dense["logdet"], rtol=1e-11)
    np.testing.assert_allclose(auto["sqmah"], dense["sqmah"], rtol=1e-8)
    b = int(np.argmax(hw))
    assert close_lnl(auto["lnl"][b], O.log_likelihood(oo, plist[b]))


def test_wide_band_halfwidth_slightly_too_small_is_flagged():
    """a half-width that is too small by a few pixels must give info = -4 in the wide (in-place) path as
    well -- the probe sits on the first diagonal past the CALLER's half-width, not past the storage width, and
    nothing beyond the caller's half-width is stored."""
    import torch

    o = synth.make_order(N=1024)
    oo = oracle_order(o)
    do = device_order(oo)
    p = synth.vector_to_oracle_params(synth.walker_ball(o, B=1)[0])
    p["global_cov"] = (p["global_cov"][0], float(np.log(25.0)))
    p["local_cov"] = []
    md, rows = pack_rows(do, [p])
    # exact support of the global kernel on this grid: largest pixel offset with r <= r0 = 6 ls (kernels.py:27-33)
    w = o["wave"]
    r0 = 6 * 25.0
    d = np.arange(1, 1024)
    r = 2.99792458e5 / 2 * (w[d] - w[0]) / (w[d] + w[0])
    support = int(d[r <= r0].max())
    assert support > do.banded_window_halfwidth()
    P = D.to_dev(rows, do.dev)
    lnl = D.empty((1,), do.dev)
    info = D.empty((1,), do.dev, torch.int32)
    dense = do.loglike(md, rows)["lnl"][0]
    for short in (1, 2, 7, 20):
        do.loglike_banded_device(md, P, support - short, lnl, info)
        assert info.cpu().numpy()[0] == D.INFO_BANDWIDTH, short
        assert lnl.cpu().numpy()[0] == -np.inf
    for extra in (0, 1, 13):
        do.loglike_banded_device(md, P, support + extra, lnl, info)
        assert info.cpu().numpy()[0] == 0, extra
        np.testing.assert_allclose(lnl.cpu().numpy()[0], dense, rtol=1e-10)
