"""Stage-level parity of the HIP kernels (through the C-ABI) against the golden vectors produced by
the reference and against the CPU oracle.  Needs an MI355X: run with -m gpu."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from starfish_amd import _lib

    return _lib.require_gpu()


@pytest.mark.parametrize("N", [64, 200])
def test_covariance_kernels_vs_reference(gpu, N):
    from starfish_amd.models.kernels import global_covariance_matrix, local_covariance_matrix

    g = load_golden("kernels.npz")
    wave = g[f"wave_{N}"]
    for i, (a, l) in enumerate(g[f"g_params_{N}"]):
        got = global_covariance_matrix(wave, a, l)
        np.testing.assert_allclose(got, g[f"g_{N}_{i}"], rtol=1e-13, atol=1e-300)
    for i, (a, mu, s) in enumerate(g[f"l_params_{N}"]):
        got = local_covariance_matrix(wave, a, mu, s)
        np.testing.assert_allclose(got, g[f"l_{N}_{i}"], rtol=1e-13, atol=1e-300)


@pytest.mark.parametrize("tag", ["s", "l"])
def test_transforms_vs_reference(gpu, tag):
    from starfish_amd import transforms as T

    g = load_golden("transforms.npz")
    grid, wave, flux = g[f"{tag}_grid"], g[f"{tag}_wave"], g[f"{tag}_flux"]
    scale = np.abs(flux).max()
    for i, v in enumerate(g[f"{tag}_vsini"]):
        got = T.rotational_broaden(grid, flux, v)
        assert np.abs(got - g[f"{tag}_rot_{i}"]).max() <= 1e-10 * scale
    for i, f in enumerate(g[f"{tag}_fwhm"]):
        got = T.instrumental_broaden(grid, flux, f)
        assert np.abs(got - g[f"{tag}_inst_{i}"]).max() <= 1e-10 * scale
    for i, vz in enumerate(g[f"{tag}_vz"]):
        sh = T.doppler_shift(grid, vz)
        np.testing.assert_array_equal(sh, g[f"{tag}_dop_{i}"])
        got = T.resample(sh, flux, g[f"{tag}_resq_{i}"])
        assert np.abs(got - g[f"{tag}_res_{i}"]).max() <= 1e-10 * scale
    got = T.chebyshev_correct(wave, g[f"{tag}_cheb_in"], g[f"{tag}_cheb_c"])
    np.testing.assert_allclose(got, g[f"{tag}_cheb"], rtol=1e-14)
    # 1-D forms
    got1 = T.rotational_broaden(grid, flux[0], g[f"{tag}_vsini"][0])
    assert got1.shape == flux[0].shape
    assert np.abs(got1 - g[f"{tag}_rot_0"][0]).max() <= 1e-10 * scale


def test_resample_irregular_grid(gpu):
    from starfish_amd import transforms as T

    g = load_golden("transforms.npz")
    got = T.resample(g["irr_x"], g["irr_y"], g["irr_q"])
    assert np.abs(got - g["irr_out"]).max() <= 1e-11


def test_transform_argument_errors(gpu):
    from starfish_amd import transforms as T

    w = np.linspace(5000, 5010, 64)
    f = np.ones(64)
    with pytest.raises(ValueError):
        T.rotational_broaden(w, f, 0.0)
    with pytest.raises(ValueError):
        T.rotational_broaden(w, f, -3.0)
    with pytest.raises(ValueError):
        T.instrumental_broaden(w, f, -1.0)
    with pytest.raises(ValueError):
        T.resample(w, f, np.array([-1.0, 5001.0]))
    with pytest.raises(ValueError):
        T.chebyshev_correct(w, f, [0.9, 0.1])
    np.testing.assert_allclose(T.instrumental_broaden(w, f, 0.0), f, atol=1e-14)


@pytest.mark.parametrize("n,batch", [(64, 3), (192, 9), (256, 5), (1024, 2), (1088, 2), (1984, 3), (640, 40), (3008, 2), (1152, 17), (2560, 28)])
def test_potrf_logdet_sqmah_random_spd(gpu, chol_sequence, n, batch):
    import torch
    from starfish_amd import _device as D, _lib

    rng = np.random.default_rng(n)
    dev = D.device_of()
    lda = n + 16
    A = np.zeros((batch, n, lda))
    R = rng.standard_normal((batch, n))
    want_ld, want_sq = [], []
    for b in range(batch):
        G = rng.standard_normal((n, n))
        S = G @ G.T / n + np.diag(rng.uniform(0.5, 2.0, n))
        A[b, :, :n] = S
        L = np.linalg.cholesky(S)
        z = np.linalg.solve(L, R[b])
        want_ld.append(2 * np.log(np.diag(L)).sum())
        want_sq.append(z @ z)
    dA = D.to_dev(A, dev)
    dR = D.to_dev(R, dev)
    info = D.empty((batch,), dev, torch.int32)
    ld = D.empty((batch,), dev)
    sq = D.empty((batch,), dev)
    ws = D.workspace(gpu.sf_potrf_workspace_bytes(n, batch), dev)
    s = D.stream_ptr(dev)
    _lib.check(gpu.sf_potrf_batch(D.ptr(dA), n, lda, n * lda, batch, D.ptr(info), D.ptr(ws), ws.numel(), s))
    _lib.check(gpu.sf_logdet_sqmah_batch(D.ptr(dA), n, lda, n * lda, batch, D.ptr(dR), n, D.ptr(ws),
                                         ws.numel(), D.ptr(ld), D.ptr(sq), s))
    assert (info.cpu().numpy() == 0).all()
    np.testing.assert_allclose(ld.cpu().numpy(), want_ld, rtol=1e-12)
    np.testing.assert_allclose(sq.cpu().numpy(), want_sq, rtol=1e-11)
    Lgpu = np.tril(dA.cpu().numpy()[0, :, :n])
    np.testing.assert_allclose(Lgpu, np.linalg.cholesky(A[0, :, :n]), rtol=0, atol=1e-12)


@pytest.mark.parametrize("n,batch,calls", [(1024, 24, 60), (1984, 5, 40), (2048, 16, 25), (2688, 28, 12)])
def test_potrf_dataflow_sequence_is_repeatable(gpu, n, batch, calls):
    """The dataflow sequence is one persistent launch whose workgroups synchronise through counters in memory: the order
    in which tasks run differs from call to call, the arithmetic must not (fixed split order of every partial sum), and
    no wait may time out (an aborted launch flags every matrix with SF_INFO_INTERNAL)."""
    import torch
    from starfish_amd import _device as D, _lib

    dev = D.device_of()
    lda = n + 16
    rng = np.random.default_rng(7)
    A = np.zeros((batch, n, lda))
    for b in range(batch):
        G = rng.standard_normal((n, 48))
        A[b, :, :n] = G @ G.T / 48 + np.diag(rng.uniform(1.0, 2.0, n))
    base = D.to_dev(A, dev)
    work = torch.empty_like(base)
    info = D.empty((batch,), dev, torch.int32)
    ws = D.workspace(gpu.sf_potrf_workspace_bytes(n, batch), dev)
    s = D.stream_ptr(dev)
    assert gpu.sf_debug_cholesky_sequence(4) == 0
    try:
        first = None
        for it in range(calls):
            work.copy_(base)
            _lib.check(gpu.sf_potrf_batch(D.ptr(work), n, lda, n * lda, batch, D.ptr(info), D.ptr(ws), ws.numel(), s))
            torch.cuda.synchronize()
            assert int(info.abs().max()) == 0, f"call {it}: info = {info.cpu().numpy().tolist()}"
            L = torch.tril(work[:, :, :n])
            if first is None:
                first = L.clone()
                err = (first[0] @ first[0].T - base[0, :, :n]).abs().max().item()
                assert err < 1e-12
            else:
                assert torch.equal(L, first), f"call {it} differs from call 0"
    finally:
        gpu.sf_debug_cholesky_sequence(-1)


def test_potrf_reports_non_positive_pivot(gpu, chol_sequence):
    import torch
    from starfish_amd import _device as D, _lib

    n, lda = 128, 144
    A = np.zeros((2, n, lda))
    A[0, :, :n] = np.eye(n)
    A[1, :, :n] = np.eye(n)
    A[1, 70, 70] = -1.0
    dev = D.device_of()
    dA = D.to_dev(A, dev)
    info = D.empty((2,), dev, torch.int32)
    ws = D.workspace(gpu.sf_potrf_workspace_bytes(n, 2), dev)
    _lib.check(gpu.sf_potrf_batch(D.ptr(dA), n, lda, n * lda, 2, D.ptr(info), D.ptr(ws), ws.numel(),
                                  D.stream_ptr(dev)))
    assert info.cpu().numpy().tolist() == [0, 71]
    # n = 64 mod 128: the fused sequences factorise in a frame shifted by 64 virtual rows (sf_potrf_front_pad); the pivot
    # index reported is the matrix's own, in the first tile (which holds the virtual rows) and after it
    n, lda = 320, 336
    A = np.zeros((3, n, lda))
    A[:, :, :n] = np.eye(n)
    A[1, 5, 5] = -1.0
    A[2, 200, 200] = -1.0
    dA = D.to_dev(A, dev)
    info = D.empty((3,), dev, torch.int32)
    ws = D.workspace(gpu.sf_potrf_workspace_bytes(n, 3), dev)
    _lib.check(gpu.sf_potrf_batch(D.ptr(dA), n, lda, n * lda, 3, D.ptr(info), D.ptr(ws), ws.numel(),
                                  D.stream_ptr(dev)))
    assert info.cpu().numpy().tolist() == [0, 6, 201]


@pytest.mark.parametrize("tag,m", [("a", 8), ("b", 4)])
def test_emulator_query_vs_reference(gpu, tag, m):
    from starfish_amd import synth
    from starfish_amd import _device as D

    g = load_golden("emulator.npz")
    o = synth.make_order(N=256, m=m, seed=3)
    do = D.DeviceOrder(
        np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0), np.zeros((0, 0)), o["grid_points"],
        g[f"{tag}_variances"], g[f"{tag}_lengthscales"], g[f"{tag}_v11"], o["w_hat"],
    )
    q = g[f"{tag}_queries"]
    mu, cov, info = do.emulator_query(q)
    assert (info == 0).all()
    for i in range(len(q)):
        np.testing.assert_allclose(mu[i], g[f"{tag}_mu_{i}"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(cov[i], g[f"{tag}_cov_{i}"], rtol=1e-9, atol=1e-9)
    _, _, info = do.emulator_query([[5999.0, 4.2, -0.3], [6050.0, 4.2, 0.01]])
    assert info.tolist() == [-1, -1]


def test_extinct_ccm89_unpinned_law(gpu):
    """extinct(): Av = 0 identity is the only case the reference's tests pin (tests/test_transforms.py:
    162-165); the CCM89 law itself is parity-unpinned and checked against the restated formulas."""
    from oracle import sf_oracle as O
    from starfish_amd import transforms as T

    w = np.linspace(1200.0, 30000.0, 777)  # far-UV .. near-IR: every branch of the law
    f = 1.0 + 0.1 * np.random.default_rng(0).standard_normal((3, 777))
    np.testing.assert_array_equal(T.extinct(w, f, 0.0), f)
    np.testing.assert_allclose(T.extinct(w, f, 0.7), O.extinct_ccm89(w, f, 0.7), rtol=1e-13)
    np.testing.assert_allclose(T.extinct(w, f[0], 1.3, Rv=4.0), O.extinct_ccm89(w, f[0], 1.3, 4.0), rtol=1e-13)
    with pytest.raises(ValueError):
        T.extinct(w, f, 1.0, law="nope")
    with pytest.raises(ValueError):
        T.extinct(w, f, 1.0, Rv=-1.0)
    # the spline-based laws (unpinned as well): restated from Fitzpatrick (1999) / Fitzpatrick & Massa (2007);
    # fm07 ignores Rv like the reference's call (transforms.py:199-200)
    np.testing.assert_allclose(T.extinct(w, f, 0.8, Rv=2.7, law="fitzpatrick99"),
                               f * 10 ** (-0.4 * O.fitzpatrick99_a_lambda(w, 0.8, 2.7)), rtol=1e-12)
    np.testing.assert_allclose(T.extinct(w, f, 1.1, law="fitzpatrick99"),
                               f * 10 ** (-0.4 * O.fitzpatrick99_a_lambda(w, 1.1)), rtol=1e-12)
    for rv in (3.1, 5.0):
        np.testing.assert_allclose(T.extinct(w, f, 0.6, Rv=rv, law="fm07"), f * 10 ** (-0.4 * O.fm07_a_lambda(w, 0.6)),
                                   rtol=1e-12)
    for law in ("fitzpatrick99", "fm07"):
        np.testing.assert_array_equal(T.extinct(w, f, 0.0, law=law), f)
        v = T.extinct(np.array([5470.0, 5500.0]), np.ones(2), 1.0, law=law)
        assert abs(-2.5 * np.log10(v[1]) - 1.0) < 3e-2  # A(V) ~ Av
        a = -2.5 * np.log10(T.extinct(w, np.ones_like(w), 1.0, law=law))
        assert np.all(np.diff(a[w > 2300]) < 0)  # monotonically falling redward of the 2175 A bump
    # the other two closed-form laws (also unpinned): restated formulas, identity at Av = 0, A(5500 A) = Av
    for law, alam in (("odonnell94", O.odonnell94_a_lambda), ("calzetti00", O.calzetti00_a_lambda)):
        np.testing.assert_array_equal(T.extinct(w, f, 0.0, law=law), f)
        np.testing.assert_allclose(T.extinct(w, f, 0.9, Rv=3.4, law=law), f * 10 ** (-0.4 * alam(w, 0.9, 3.4)),
                                   rtol=1e-13)
        v = T.extinct(np.array([5494.5, 5500.0]), np.ones(2), 1.0, Rv=3.1, law=law)
        assert abs(-2.5 * np.log10(v[1]) - 1.0) < 2e-3


def test_extinct_kernels_reproduce_the_papers_tables(gpu):
    """The DEVICE laws against numbers printed in the papers (no oracle in between): Fitzpatrick (1999) Table 3,
    Calzetti et al. (2000) eq. 4 by hand, Fitzpatrick & Massa (2007) anchors, CCM89 Table 3.  extinct() stays
    parity-unpinned (third-party `extinction` absent); this is the strongest pin available."""
    from starfish_amd import transforms as T

    def a_lambda(w, law, **kw):  # A_lambda / Av through the kernel
        w = np.asarray(w, dtype=float)
        return -2.5 * np.log10(T.extinct(w, np.ones_like(w), 1.0, law=law, **kw))

    lam = np.array([26500.0, 12200.0, 6000.0, 5470.0, 4670.0, 4110.0, 2700.0, 2600.0])
    np.testing.assert_allclose(3.1 * a_lambda(lam, "fitzpatrick99", Rv=3.1),
                               [0.265, 0.829, 2.688, 3.055, 3.806, 4.315, 6.265, 6.591], atol=1.5e-3)
    np.testing.assert_allclose(4.05 * a_lambda([5500.0, 22000.0, 1200.0], "calzetti00", Rv=4.05), [4.05, 0.3692, 12.119], atol=5e-3)
    np.testing.assert_allclose(3.1 * (a_lambda([5530.0, 4000.0, 3300.0], "fm07") - 1.0), [0.0, 1.322, 2.055], atol=1e-9)
    band_x = np.array([2.78, 1.82, 1.43, 1.11, 0.80])  # U V R I J of CCM89 Table 3 (Rv = 3.1)
    np.testing.assert_allclose(a_lambda(1e4 / band_x, "ccm89", Rv=3.1), [1.569, 1.000, 0.751, 0.479, 0.282], atol=1.5e-3)
    a1 = 1 + 0.104 - 0.609 + 0.701 + 1.137 - 1.718 - 0.827 + 1.647 - 0.505  # O'Donnell (1994) coefficient sums at y = 1
    b1 = 1.952 + 2.908 - 3.989 - 7.985 + 11.102 + 5.491 - 10.805 + 3.347
    np.testing.assert_allclose(a_lambda([1e4 / 2.82, 1e4 / 1.82], "odonnell94", Rv=3.1), [a1 + b1 / 3.1, 1.0], atol=1e-12)
