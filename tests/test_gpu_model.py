"""End-to-end parity of the batched HIP path (transform chain, fused fill, Cholesky, solve) against
the CPU oracle and the reference-generated golden vectors.  Needs an MI355X: run with -m gpu."""
import os
import sys

import numpy as np
import pytest

from conftest import load_golden
from oracle import sf_oracle as O
from starfish_amd import synth

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import gen_golden_cases as G  # noqa: E402
from gpu_helpers import device_order, oracle_order, pack_rows  # noqa: E402

pytestmark = pytest.mark.gpu

LNL_RTOL = 1e-8  # |dlnL| <= 1e-8 |lnL| + 1e-8  


def close_lnl(got, want):
    return abs(got - want) <= LNL_RTOL * abs(want) + 1e-8


def small_params(o, name, factors):
    from scipy.interpolate import LinearNDInterpolator

    c = G.small_case_params(o, G.SMALL_CASES[name])
    p = dict(grid=c["grid_params"])
    for k in ("vz", "vsini", "log_scale", "cheb"):
        if k in c:
            p[k] = c[k]
    if "global_cov" in c:
        p["global_cov"] = (c["global_cov"]["log_amp"], c["global_cov"]["log_ls"])
    if "local_cov" in c:
        p["local_cov"] = [(k["mu"], k["log_amp"], k["log_sigma"]) for k in c["local_cov"]]
    if G.SMALL_CASES[name].get("norm"):
        p["norm"] = float(LinearNDInterpolator(o["grid_points"], factors, rescale=True)(np.asarray(p["grid"])))
    return p


@pytest.fixture(scope="module")
def small():
    o = synth.make_order(N=256, m=4, seed=5)
    oo = oracle_order(o)
    return o, oo, device_order(oo)


@pytest.mark.parametrize("name", list(G.SMALL_CASES))
def test_small_model_cases_vs_reference(small, name, chol_sequence):
    o, oo, do = small
    g = load_golden("model_small.npz")
    p = small_params(o, name, g["factors"])
    md, rows = pack_rows(do, [p])
    ref = g[f"{name}_lnl"]

    tr = do.transform(md, rows)
    assert tr["info"][0] == 0
    f_or, c_or, _ = O.forward_model(oo, p)
    np.testing.assert_allclose(tr["flux"][0], g[f"{name}_flux"], rtol=0, atol=1e-10 * np.abs(f_or).max())
    assert abs(tr["log_scale"][0] - ref[3]) <= 1e-10 * max(1.0, abs(ref[3]))
    np.testing.assert_allclose(tr["resid"][0], g[f"{name}_flux"] - oo.flux, rtol=0, atol=1e-10)

    fw = do.forward(md, rows)
    atol = 1e-11 * np.abs(c_or).max()
    if f"{name}_cov" in g:
        np.testing.assert_allclose(fw["cov"][0], g[f"{name}_cov"], rtol=1e-10, atol=atol)
    else:
        np.testing.assert_allclose(fw["cov"][0][G.COV_ROWS], g[f"{name}_covrows"], rtol=1e-10, atol=atol)
        np.testing.assert_allclose(fw["cov"][0].diagonal(), g[f"{name}_diag"], rtol=1e-10)
    np.testing.assert_allclose(fw["cov"][0], fw["cov"][0].T, rtol=0, atol=0)

    ll = do.loglike(md, rows)
    assert ll["info"][0] == 0
    assert close_lnl(ll["lnl"][0], ref[0])
    assert abs(ll["logdet"][0] - ref[1]) <= 1e-10 * abs(ref[1])
    assert abs(ll["sqmah"][0] - ref[2]) <= 1e-8 * abs(ref[2])


def test_batch_matches_oracle_and_reference_n1024(chol_sequence):
    g = load_golden("model_large.npz")
    o = synth.make_order(N=1024)
    oo = oracle_order(o)
    do = device_order(oo)
    P = synth.walker_ball(o, B=128)
    plist = [synth.vector_to_oracle_params(p) for p in P[:16]]
    md, rows = pack_rows(do, plist)
    out = do.loglike(md, rows, want_resid=True)
    assert (out["info"] == 0).all()
    for b in range(8):  # from the real reference
        assert close_lnl(out["lnl"][b], g["n1024_batch_lnl"][b])
    for b in (8, 15):  # beyond the fixture: the pinned oracle
        want, logdet, sqmah, R = O.log_likelihood(oo, plist[b], return_parts=True)
        assert close_lnl(out["lnl"][b], want)
        np.testing.assert_allclose(out["resid"][b], R, rtol=0, atol=1e-10)
    # centre point: the survey's known answer
    md, rows = pack_rows(do, [synth.vector_to_oracle_params(synth.centre_vector(o))])
    assert abs(do.loglike(md, rows)["lnl"][0] - 4124.8909586559) < 1e-6
    # chunked evaluation gives identical numbers
    md, rows = pack_rows(do, plist)
    out2 = do.loglike(md, rows, max_chunk=5)
    np.testing.assert_array_equal(out2["lnl"], out["lnl"])


def test_ragged_size_n3000_and_sampled_covariance(chol_sequence):
    g = load_golden("model_large.npz")
    o = synth.make_order(N=3000)
    oo = oracle_order(o)
    do = device_order(oo)
    P = g["n3000_batch_P"]
    plist = [synth.vector_to_oracle_params(synth.centre_vector(o))] + [synth.vector_to_oracle_params(p) for p in P]
    md, rows = pack_rows(do, plist)
    out = do.loglike(md, rows)
    assert close_lnl(out["lnl"][0], g["n3000_lnl"][0])
    for b in range(len(P)):
        assert close_lnl(out["lnl"][1 + b], g["n3000_batch_lnl"][b])
    fw = do.forward(md, rows[:1])
    cov = fw["cov"][0]
    np.testing.assert_allclose(fw["flux"][0], g["n3000_flux"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(cov.diagonal(), g["n3000_diag"], rtol=1e-10)
    np.testing.assert_allclose(cov[g["n3000_ii"], g["n3000_jj"]], g["n3000_vals"], rtol=1e-10,
                               atol=1e-11 * np.abs(g["n3000_diag"]).max())
    np.testing.assert_allclose(cov.sum(axis=1), g["n3000_rowsum"], rtol=1e-9, atol=1e-12)


def test_out_of_grid_and_bad_vsini_are_flagged_not_fatal():
    o = synth.make_order(N=256, m=4, seed=5)
    oo = oracle_order(o)
    do = device_order(oo)
    good = synth.vector_to_oracle_params(synth.centre_vector(o))
    bad_grid = dict(good, grid=[5990.0, 4.2, -0.3])
    bad_vsini = dict(good, vsini=-1.0)
    md, rows = pack_rows(do, [good, bad_grid, bad_vsini, good])
    out = do.loglike(md, rows)
    assert out["info"].tolist() == [0, -1, -2, 0]
    assert np.isfinite(out["lnl"][0]) and out["lnl"][0] == out["lnl"][3]
    assert out["lnl"][1] == -np.inf and out["lnl"][2] == -np.inf


def test_wasp14_order_plumbing(chol_sequence):
    """BASELINE config 1: bundled WASP14 order 23 (masked, N = 1932) against the reference value."""
    d = load_golden("wasp14_order23.npz")
    g = load_golden("model_wasp14.npz")
    mask = d["mask"]
    oo = O.OracleOrder(d["wave"][mask], d["flux"][mask], d["sigma"][mask], g["emu_wl"], g["eigenspectra"],
                       g["flux_mean"], g["flux_std"], g["grid_points"], g["w_hat"])
    do = device_order(oo)
    vec = g["vector"]
    assert tuple(g["labels"]) == synth.LABELS
    p = synth.vector_to_oracle_params(vec)
    md, rows = pack_rows(do, [p])
    out = do.loglike(md, rows)
    assert out["info"][0] == 0
    assert close_lnl(out["lnl"][0], g["lnl"][0])
    assert close_lnl(O.log_likelihood(oo, p), g["lnl"][0])


def test_cfg2_full_size_n4096_vs_reference(chol_sequence):
    """BASELINE config 2 at full size: the reference's own values (tools/gen_golden.py --big)."""
    g = load_golden("model_cfg2.npz")
    o = synth.make_order(N=4096)
    oo = oracle_order(o)
    do = device_order(oo)
    P = g["n4096_batch_P"]
    plist = [synth.vector_to_oracle_params(synth.centre_vector(o))] + [synth.vector_to_oracle_params(p) for p in P]
    md, rows = pack_rows(do, plist)
    out = do.loglike(md, rows)
    assert (out["info"] == 0).all()
    assert close_lnl(out["lnl"][0], g["n4096_lnl"][0])
    assert abs(out["lnl"][0] - 16579.1706341206) < 1e-6  # SURVEY.md section 8c known answer
    assert abs(out["logdet"][0] - g["n4096_lnl"][1]) <= 1e-10 * abs(g["n4096_lnl"][1])
    assert abs(out["sqmah"][0] - g["n4096_lnl"][2]) <= 1e-8 * abs(g["n4096_lnl"][2])
    for b in range(len(P)):
        assert close_lnl(out["lnl"][1 + b], g["n4096_batch_lnl"][b])
    # size-independent properties at full size: replicas agree bit for bit, order in the batch is irrelevant
    # (across DIFFERENT batch sizes the split-K factor of the late panels may differ: same value to rounding only)
    rep = do.loglike(md, np.repeat(rows[:2], 3, axis=0))
    assert rep["lnl"][0] == rep["lnl"][1] == rep["lnl"][2]
    assert rep["lnl"][3] == rep["lnl"][4] == rep["lnl"][5]
    np.testing.assert_allclose(rep["lnl"][[0, 3]], out["lnl"][:2], rtol=1e-12)
    fw = do.forward(md, rows[:1])
    cov = fw["cov"][0]
    np.testing.assert_allclose(cov.diagonal(), g["n4096_diag"], rtol=1e-10)
    np.testing.assert_allclose(cov[g["n4096_ii"], g["n4096_jj"]], g["n4096_vals"], rtol=1e-10,
                               atol=1e-11 * np.abs(g["n4096_diag"]).max())
    np.testing.assert_allclose(fw["flux"][0], g["n4096_flux"], rtol=0, atol=1e-10)


@pytest.mark.parametrize("batch", [32, 45])
def test_cfg2_strong_split_batches_dataflow_vs_fused_and_oracle(batch):
    """The batches a strong split of config 2 leaves per GPU run on the dataflow sequence by the library's own choice; from
    21 to 48 matrices its front widens over the last panels (per-panel front width, counters that start late).  Same walkers
    through the fused launch sequence: same values to rounding; two of them against the pinned oracle; the reference's own
    values for the walkers the fixture holds."""
    from starfish_amd import _lib

    g = load_golden("model_cfg2.npz")
    o = synth.make_order(N=4096)
    oo = oracle_order(o)
    do = device_order(oo)
    P = synth.walker_ball(o, B=batch, seed=3)
    nref = min(len(g["n4096_batch_P"]), 4)
    plist = [synth.vector_to_oracle_params(p) for p in g["n4096_batch_P"][:nref]] + [synth.vector_to_oracle_params(p) for p in P[nref:]]
    md, rows = pack_rows(do, plist)
    lib = _lib.require_gpu()
    auto = do.loglike(md, rows)  # (batch x panels <= 2048: dataflow)
    assert (auto["info"] == 0).all()
    assert lib.sf_debug_cholesky_sequence(0) == 0
    try:
        fused = do.loglike(md, rows)
    finally:
        lib.sf_debug_cholesky_sequence(-1)
    assert (fused["info"] == 0).all()
    np.testing.assert_allclose(auto["lnl"], fused["lnl"], rtol=1e-12)
    np.testing.assert_allclose(auto["logdet"], fused["logdet"], rtol=1e-13)
    np.testing.assert_allclose(auto["sqmah"], fused["sqmah"], rtol=1e-10)
    for b in range(nref):
        assert close_lnl(auto["lnl"][b], g["n4096_batch_lnl"][b])
    for b in (nref, batch - 1):
        assert close_lnl(auto["lnl"][b], O.log_likelihood(oo, plist[b]))
    again = do.loglike(md, rows)
    np.testing.assert_array_equal(again["lnl"], auto["lnl"])


def test_cfg5_long_order_n16384_vs_reference(chol_sequence):
    """BASELINE config 5 (N = 16384, N_f = 32768: FFT through the global-memory path) against the value
    produced by the real reference."""
    g = load_golden("model_cfg5.npz")
    o = synth.make_order(N=16384)
    oo = oracle_order(o)
    do = device_order(oo)
    assert do.nf == 32768
    p = synth.vector_to_oracle_params(synth.centre_vector(o))
    md, rows = pack_rows(do, [p, p])
    out = do.loglike(md, rows, want_resid=True)
    assert (out["info"] == 0).all()
    ref = g["n16384_lnl"]
    assert close_lnl(out["lnl"][0], ref[0]) and out["lnl"][0] == out["lnl"][1]
    assert abs(out["lnl"][0] - 66263.72856408486) < 1e-5
    assert abs(out["logdet"][0] - ref[1]) <= 1e-10 * abs(ref[1])
    assert abs(out["sqmah"][0] - ref[2]) <= 1e-8 * abs(ref[2])
    np.testing.assert_allclose(out["resid"][0], g["n16384_flux"] - oo.flux, rtol=0, atol=1e-10)


def test_model_with_extinction_matches_oracle_unpinned():
    """Av in the model (reference test fixture uses Av=0, tests/conftest.py:123): Av = 0 reproduces the
    reference value exactly; Av != 0 is checked against the (parity-unpinned) oracle restatement."""
    g = load_golden("model_small.npz")
    o = synth.make_order(N=256, m=4, seed=5)
    oo = oracle_order(o)
    do = device_order(oo)
    base = small_params(o, "full", g["factors"])
    p0 = dict(base, Av=0.0)
    p1 = dict(base, Av=0.35)
    md, rows = pack_rows(do, [p0, p1])
    out = do.loglike(md, rows)
    assert (out["info"] == 0).all()
    assert close_lnl(out["lnl"][0], g["full_lnl"][0])
    assert close_lnl(out["lnl"][1], O.log_likelihood(oo, p1))
    assert out["lnl"][0] != out["lnl"][1]


@pytest.mark.parametrize("N", [70, 200, 300, 520])
def test_small_and_ragged_sizes_both_solvers(N, chol_sequence):
    """Orders shorter than one panel, exactly one panel, and with a narrower last panel (padding to 64 for the
    dense factorisation, to 16 for the banded one): both solvers against the oracle."""
    o = synth.make_order(N=N, m=4, seed=3)
    oo = oracle_order(o)
    do = device_order(oo)
    plist = [synth.vector_to_oracle_params(p) for p in synth.walker_ball(o, B=5, seed=2)]
    md, rows = pack_rows(do, plist)
    dense = do.loglike(md, rows)
    auto = do.loglike(md, rows, solver="auto")
    assert (dense["info"] == 0).all() and (auto["info"] == 0).all()
    for b, p in enumerate(plist):
        want = O.log_likelihood(oo, p)
        assert close_lnl(dense["lnl"][b], want) and close_lnl(auto["lnl"][b], want)


def test_perturbed_grid_full_size_vs_oracle(chol_sequence):
    """cfg 2 size on a NON log-uniform wavelength grid: the fill evaluates K_global entry by entry (no per-diagonal
    table), through the product API, against the oracle."""
    o = synth.perturb_grid(synth.make_order(N=4096))
    rel = np.diff(o["wave"]) / o["wave"][:-1]
    assert rel.max() / rel.min() > 1.05  # really not log-uniform
    model = synth.build_model(o)
    P = synth.walker_ball(o, B=2, seed=11)
    got, info = model.log_likelihood_batch(P, return_info=True)
    assert (info == 0).all()
    oo = oracle_order(o)
    for b in range(2):
        want = O.log_likelihood(oo, synth.vector_to_oracle_params(P[b]))
        assert close_lnl(got[b], want), (got[b], want)


def test_full_size_properties_permutation_and_null_kernel(chol_sequence):
    """Size-independent properties at cfg-2 size, dense and structured solver: the order of the walkers in a batch is
    irrelevant (bit for bit: every walker is its own matrix), and a local kernel of zero amplitude changes nothing."""
    o = synth.make_order(N=4096)
    do = device_order(oracle_order(o))
    plist = [synth.vector_to_oracle_params(p) for p in synth.walker_ball(o, B=6, seed=5)]
    md, rows = pack_rows(do, plist)
    perm = np.array([3, 0, 5, 1, 4, 2])
    for solver in ("dense", "auto"):
        a = do.loglike(md, rows, solver=solver)
        b = do.loglike(md, rows[perm], solver=solver)
        assert (a["info"] == 0).all()
        np.testing.assert_array_equal(b["lnl"], a["lnl"][perm])
        np.testing.assert_array_equal(b["logdet"], a["logdet"][perm])
    # exp(-800) == 0.0: the patch of the second local kernel is filled with exact zeros
    base = do.loglike(md, rows)
    ghost = [dict(p, local_cov=list(p["local_cov"]) + [(float(o["wave"][2500]), -800.0, float(np.log(12.0)))]) for p in plist]
    md2, rows2 = pack_rows(do, ghost)
    for solver in ("dense", "auto"):
        g = do.loglike(md2, rows2, solver=solver)
        assert (g["info"] == 0).all()
        np.testing.assert_allclose(g["lnl"], base["lnl"], rtol=1e-13)


def test_cov_fill_entry_point_matches_forward_and_reference():
    """sf_cov_fill_batch (a3 + a4 + a14 + a15 fused, on its own): the same matrix as SpectrumModel.__call__
    (reference golden), into a caller-chosen layout (padded rows, lower triangle only) and with the likelihood's
    1e-10 jitter on request."""
    g = load_golden("model_small.npz")
    o = synth.make_order(N=256, m=4, seed=5)
    oo = oracle_order(o)
    do = device_order(oo)
    p = small_params(o, "full", g["factors"])
    md, rows = pack_rows(do, [p, p])
    fw = do.forward(md, rows)["cov"][0]
    full, info = do.cov_fill(md, rows)
    assert (info == 0).all() and full.shape == (2, 256, 256)
    np.testing.assert_array_equal(full[0], fw)            # the same kernel, the same bits
    np.testing.assert_array_equal(full[0], full[0].T)
    np.testing.assert_array_equal(full[0], full[1])
    np.testing.assert_allclose(full[0], g["full_cov"], rtol=1e-10, atol=1e-11 * np.abs(g["full_cov"]).max())
    padded, _ = do.cov_fill(md, rows[:1], ld=272, lower_only=True, add_jitter=True)
    assert padded.shape == (1, 256, 272)
    lo = np.tril_indices(256)
    want = fw.copy()
    want[np.diag_indices(256)] += 1e-10                   # spectrum_model.py:399
    np.testing.assert_array_equal(padded[0][:, :256][lo], want[lo])
    # untouched: the padding columns and everything a whole tile above the diagonal (inside the tiles on the diagonal
    # the upper part may be written -- the factorisation never reads it)
    ii, jj = np.triu_indices(256, 128)
    assert not padded[0][:, 256:].any() and not padded[0][:, :256][ii, jj].any()


@pytest.mark.parametrize("N,ld", [(3000, 3000), (250, 251)])
def test_cov_fill_writes_only_the_callers_rows(N, ld):
    """sf_cov_fill_batch into caller matrices of exactly n rows when n is not a multiple of 64 (cfg 3: 3000 -> the
    workspace layout pads to 3008 with an identity block; a caller's array has no such rows): nothing lands in the
    next matrix or behind the last one, lower triangle = the full fill's, with ld = n and with an odd ld."""
    o = synth.make_order(N=N, m=4, seed=9)
    do = device_order(oracle_order(o))
    plist = [synth.vector_to_oracle_params(p) for p in synth.walker_ball(o, B=2, seed=4)]
    md, rows = pack_rows(do, plist)
    full, info = do.cov_fill(md, rows, add_jitter=True)
    assert (info == 0).all()
    for b in range(2):  # symmetric bit for bit (the dense fill writes the structured tiles above the diagonal as mirror images)
        np.testing.assert_array_equal(full[b], full[b].T)
    low, info, guard = do.cov_fill(md, rows, ld=ld, lower_only=True, add_jitter=True, guard=64 * ld + 4096)
    assert (info == 0).all() and low.shape == (2, N, ld)
    assert (guard == -7.0).all()                       # nothing behind the last matrix
    lo = np.tril_indices(N)
    for b in range(2):
        np.testing.assert_array_equal(low[b][:, :N][lo], full[b][lo])
        assert not low[b][:, N:].any()                 # padding columns untouched
        ii, jj = np.triu_indices(N, 128)
        assert not low[b][:, :N][ii, jj].any()         # nothing a whole tile above the diagonal
    # the first rows of matrix 1 hold matrix 1's own values (the identity padding of matrix 0 used to land here)
    np.testing.assert_array_equal(np.tril(low[1][:64, :64]), np.tril(full[1][:64, :64]))


def _zero_noise_model(N):
    """cfg-2 style order whose data carry NO pixel noise (sigma = 0): C = Y^T Y + K_global + K_local + 1e-10 I.  With a
    calibration ``log_scale`` of 18 the rank-m term is ~1e15 times larger than everything else and its rounding noise
    swamps the rest: C is numerically singular and numpy / LAPACK stop at the (m+1)-th pivot (checked on the oracle:
    "9-th leading minor of the array is not positive definite") -- finite parameters, the case the reference meets at
    spectrum_model.py:400."""
    o = dict(synth.make_order(N=N))
    o["sigma"] = np.zeros(N)
    return o, synth.build_model(o)


@pytest.mark.parametrize("chol_sequence", ["wide", "fused"], indirect=True)
def test_non_positive_definite_walkers_in_a_full_batch(chol_sequence):
    """SURVEY 5 / 8(b), reference spectrum_model.py:400 (cho_factor raises LinAlgError): in a batch the walkers whose
    covariance is not positive definite come back as -inf with info = LAPACK's pivot index, every other walker is
    bit-identical to the same batch without them; the scalar API raises numpy.linalg.LinAlgError like the reference."""
    N, B, bad = 4096, 128, (5, 127)
    o, model = _zero_noise_model(N)
    P = synth.walker_ball(o, B=B, seed=21)
    Pbad = P.copy()
    Pbad[list(bad), 2] = 18.0
    good, info0 = model.log_likelihood_batch(P, return_info=True)
    assert (info0 == 0).all() and np.isfinite(good).all()
    got, info = model.log_likelihood_batch(Pbad, return_info=True)
    m = 8
    for b in range(B):
        if b in bad:
            assert got[b] == -np.inf and m < info[b] <= m + 8, (b, got[b], info[b])  # oracle / LAPACK: the 9-th minor
        else:
            assert info[b] == 0 and got[b] == good[b], (b, got[b], good[b])
    oo = oracle_order(o)
    for b in (0, 126):
        assert close_lnl(got[b], O.log_likelihood(oo, synth.vector_to_oracle_params(P[b])))
    with pytest.raises(np.linalg.LinAlgError):
        O.log_likelihood(oo, synth.vector_to_oracle_params(Pbad[5]))
    # scalar API: raises like the reference; the model stays usable afterwards
    model.set_param_vector(Pbad[5])
    with pytest.raises(np.linalg.LinAlgError, match="leading minor"):
        model.log_likelihood()
    model.set_param_vector(P[5])
    assert model.log_likelihood() == pytest.approx(good[5], rel=1e-9)  # (B = 1 takes another launch sequence; C is ill-conditioned here)


@pytest.mark.parametrize("chol_sequence", ["wide", "fused"], indirect=True)
def test_poisoned_covariance_buffers_report_lapack_pivot_indices(chol_sequence):
    """sf_cov_fill_batch -> sf_potrf_batch at cfg 2's full size (128 x 4096): two matrices get a negative diagonal
    entry at a known position -> info = that position + 1 (LAPACK's convention) for those two, 0 and bit-identical
    factors for the rest."""
    import torch
    from starfish_amd import _device as D, _lib

    gpu = _lib.require_gpu()
    N, B = 4096, 128
    lda = N + 16
    o = synth.make_order(N=N)
    do = device_order(oracle_order(o))
    plist = [synth.vector_to_oracle_params(p) for p in synth.walker_ball(o, B=B, seed=3)]
    md, rows = pack_rows(do, plist)
    dev = do.dev
    with torch.cuda.device(dev):
        P = D.to_dev(rows, dev)
        cov = torch.zeros((B, N, lda), dtype=torch.float64, device=dev)
        info = D.empty((B,), dev, torch.int32)
        ws = do._work(md, B)
        s = D.stream_ptr(dev)
        _lib.check(gpu.sf_cov_fill_batch(do.ctx, C_byref(md), B, D.ptr(P), D.ptr(cov), lda, N * lda, 1, 1, D.ptr(info),
                                         D.ptr(ws), ws.numel(), s))
        assert (info.cpu().numpy() == 0).all()
        clean = cov.clone()
        pw = D.workspace(gpu.sf_potrf_workspace_bytes(N, B), dev)
        _lib.check(gpu.sf_potrf_batch(D.ptr(clean), N, lda, N * lda, B, D.ptr(info), D.ptr(pw), pw.numel(), s))
        assert (info.cpu().numpy() == 0).all()
        cov[5, 1000, 1000] = -1.0
        cov[127, N - 1, N - 1] = -1.0
        _lib.check(gpu.sf_potrf_batch(D.ptr(cov), N, lda, N * lda, B, D.ptr(info), D.ptr(pw), pw.numel(), s))
        got = info.cpu().numpy()
        want = np.zeros(B, dtype=got.dtype)
        want[5], want[127] = 1001, N
        np.testing.assert_array_equal(got, want)
        keep = [b for b in range(B) if b not in (5, 127)]
        tri = torch.tril(torch.ones((N, N), dtype=torch.bool, device=dev))
        for b in keep[::9] + [4, 6, 126]:
            assert torch.equal(cov[b, :, :N][tri], clean[b, :, :N][tri]), b
        # matrix 5: everything left of the failing column is still the factor
        assert torch.equal(torch.tril(cov[5, :1000, :1000]), torch.tril(clean[5, :1000, :1000]))


def C_byref(md):
    import ctypes

    return ctypes.byref(md)


def test_paper_form_of_the_emulator_covariance_is_a_non_default_switch():
    """the reference's CODE applies the inverse weight covariance, X^T Sigma_w^-1 X
    (spectrum_model.py:334-335) -- the default and the parity target; its docs print Phi Sigma_w Phi^T
    (docs/api/emulator.rst:105).  ``emulator_cov="paper"`` evaluates that form; checked against the oracle's restatement of
    it (no reference value exists: the reference never computes it), the default against the reference golden."""
    g = load_golden("model_small.npz")
    o = synth.make_order(N=256, m=4, seed=5)
    oo = oracle_order(o)
    do = device_order(oo)
    base = small_params(o, "full", g["factors"])
    pp = dict(base, emulator_cov="paper")
    md, rows = pack_rows(do, [pp])
    out = do.loglike(md, rows)
    fw = do.forward(md, rows)
    _, c_or, _ = O.forward_model(oo, pp)
    np.testing.assert_allclose(fw["cov"][0], c_or, rtol=1e-10, atol=1e-11 * np.abs(c_or).max())
    assert out["info"][0] == 0 and close_lnl(out["lnl"][0], O.log_likelihood(oo, pp))
    md0, rows0 = pack_rows(do, [base])
    assert close_lnl(do.loglike(md0, rows0)["lnl"][0], g["full_lnl"][0])
    assert abs(out["lnl"][0] - g["full_lnl"][0]) > 1e-4  # (the two forms really differ: 1e-3 here, far above the 1e-8 tolerance)
    # through the product API
    m_code = synth.build_model(o)
    m_paper = synth.build_model(o, emulator_cov="paper")
    assert m_code.emulator_cov == "code"
    P = synth.walker_ball(o, B=2, seed=3)
    lp, lc = m_paper.log_likelihood_batch(P), m_code.log_likelihood_batch(P)
    for b in range(2):
        p = dict(synth.vector_to_oracle_params(P[b]))
        assert close_lnl(lc[b], O.log_likelihood(oo, p))
        assert close_lnl(lp[b], O.log_likelihood(oo, dict(p, emulator_cov="paper")))


@pytest.mark.parametrize("N,ld,m,structured", [(3000, 3000, 8, True), (3000, 3000, 8, False), (1000, 1000, 4, False),
                                               (1001, 1016, 12, True), (1001, 1016, 16, False), (2000, 2008, 8, True)])
def test_dense_fill_on_rows_that_alternate_between_two_line_phases(N, ld, m, structured):
    """Row stride = 8 mod 16 doubles (cfg 3's N = 3000 with ld = N): every other row starts 64 bytes into a 128-byte line.  The
    matrix must equal the line-aligned layout's (ld a multiple of 16) BIT FOR BIT -- three matrices, so that with an odd number
    of rows the phase also flips from matrix to matrix; ranks m = 4 ... 16 (one to four MFMA K steps), with and without
    structured kernels -- and nothing may land in the padding columns or behind the last matrix.  (Written for a kernel with a
    per-row-parity column window, profiles/r05_l_fill_shifted_rows_ab.txt: measured, not kept; the layout property stays
    tested.)"""
    o = synth.make_order(N=N, m=m, seed=9)
    do = device_order(oracle_order(o))
    plist = [synth.vector_to_oracle_params(p) for p in synth.walker_ball(o, B=3, seed=4)]
    if not structured:
        plist = [{k: v for k, v in p.items() if k not in ("global_cov", "local_cov")} for p in plist]
    md, rows = pack_rows(do, plist)
    ref, info = do.cov_fill(md, rows, ld=-(-N // 16) * 16 + 16, add_jitter=True)
    assert (info == 0).all()
    got, info, guard = do.cov_fill(md, rows, ld=ld, add_jitter=True, guard=4096)
    assert (info == 0).all() and got.shape == (3, N, ld)
    assert (guard == -7.0).all()
    for b in range(3):
        np.testing.assert_array_equal(got[b][:, :N], ref[b][:, :N])
        np.testing.assert_array_equal(got[b][:, :N], got[b][:, :N].T)
        assert not got[b][:, N:].any()
    f_or, c_or, _ = O.forward_model(oracle_order(o), plist[1])
    c_or[np.diag_indices(N)] += 1e-10
    np.testing.assert_allclose(got[1][:, :N], c_or, rtol=1e-10, atol=1e-11 * np.abs(c_or).max())
