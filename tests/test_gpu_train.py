"""SpectrumModel.train with the simplex evaluated in batches (starfish_amd/_neldermead.py) on the GPU: the iterate sequence
of scipy.optimize.minimize(method="Nelder-Mead") over the scalar log-likelihood -- what the reference runs at
Starfish/models/spectrum_model.py:685-692 -- on the small golden case, and the model state the run leaves behind.
Run with -m gpu.  (tests/test_neldermead.py holds the method itself to scipy bit for bit on the CPU.)"""
import time

import numpy as np
import pytest
import scipy.stats as st
from scipy.optimize import minimize

from starfish_amd import synth

pytestmark = pytest.mark.gpu

PRIORS = {"global_cov:log_amp": st.norm(-9, 5), "vsini": st.uniform(0, 500), "T": st.uniform(6000, 200)}


def _model(N=256, m=4, seed=5):
    return synth.build_model(synth.make_order(N=N, m=m, seed=seed))


def test_batched_train_follows_scipys_iterates_on_the_small_golden_case():
    """Same start, same options: the best vertex after every iteration (`allvecs`), nit, nfev and the final simplex of
    the batched run equal those of scipy driving the scalar likelihood.  The function values differ in the last bits (a
    batch of four takes a different summation order in the Cholesky than a batch of one: equal to ~1e-13 relative), so the
    comparison is to rounding, not bit for bit: near ties could in principle order differently, on this case none does."""
    opts = dict(maxiter=60, return_all=True)
    ref = _model()
    x0 = ref.get_param_vector()

    def nll(P):
        ref.set_param_vector(P)
        return -ref.log_likelihood(PRIORS)

    want = minimize(nll, x0, method="Nelder-Mead", options=opts)
    model = _model()
    got = model.train(PRIORS, options=opts)
    assert (got.nit, got.nfev, got.status) == (want.nit, want.nfev, want.status)
    assert len(got.allvecs) == len(want.allvecs) == want.nit  # (scipy counts the initial simplex as iteration 1)
    for a, b in zip(got.allvecs, want.allvecs):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(got.final_simplex[1], want.final_simplex[1], rtol=1e-10)
    np.testing.assert_allclose(got.x, want.x, rtol=1e-9)
    # 14 vertices in one call, then four candidates per iteration (+ shrinks): far fewer device calls than evaluations
    assert got.nbatches <= got.nit + 2 and got.nfev_speculative >= got.nfev
    # the state the reference's loop leaves: last objective call's point, residual + _lnprob of that call (not success: maxiter)
    assert not got.success
    np.testing.assert_allclose(model.get_param_vector(), ref.get_param_vector(), rtol=1e-9)
    assert model._lnprob == pytest.approx(ref._lnprob, rel=1e-10)
    assert len(model.residuals) == 1
    # the serial path is still there and gives scipy's own result
    serial = _model()
    s = serial.train(PRIORS, batch_simplex=False, options=dict(maxiter=60))
    np.testing.assert_array_equal(s.x, want.x)
    assert 1 < len(serial.residuals) <= want.nfev  # (one entry per evaluation that reached the device: not those the prior rejects)


def test_converged_run_sets_the_solution_and_invalid_used_points_raise_like_the_scalar_objective():
    model = _model()
    model.freeze(["global_cov", "local_cov", "cheb", "Z", "logg", "vz", "log_scale"])  # (T, vsini): a quick convergence
    labels = model.labels
    soln = model.train(PRIORS, options=dict(xatol=1e-3, fatol=1e-3))
    assert soln.success and soln.nit > 5
    np.testing.assert_array_equal(model.get_param_vector(), soln.x)
    assert model.labels == labels
    # a start at the edge of the emulator grid: the reflection leaves it, scipy's objective would raise ValueError there
    edge = _model()
    edge.freeze(["global_cov", "local_cov", "cheb", "Z", "logg", "vz", "log_scale", "vsini"])
    edge["T"] = 6199.0
    with pytest.raises(ValueError, match="outside of original parameter range"):
        edge.train(options=dict(initial_simplex=[[6199.0], [6180.0]], maxiter=4))


def test_batched_simplex_is_faster_per_iteration_than_the_serial_loop_at_n4096():
    """The point of it: at N = 4096 a scalar evaluation is a latency-bound launch of one matrix (~3.8 ms); four
    candidates in one batch cost ~1.2 x that.  Timed over the same 12 iterations, initial simplex included."""
    o = synth.make_order(N=4096)
    a, b = synth.build_model(o), synth.build_model(o)
    for m in (a, b):
        m.log_likelihood()  # (contexts, workspaces, first launches)
        m.log_likelihood_batch(np.tile(m.get_param_vector(), (14, 1)))
    opts = dict(maxiter=12)
    # (scipy's default simplex enlarges every coordinate by 5 %: T = 6050 K -> 6352 K leaves the emulator grid, where the
    # objective raises -- in the reference too; a prior keeps such vertices at -inf without a device call)
    t0 = time.perf_counter()
    s1 = a.train(PRIORS, batch_simplex=False, options=opts)
    t_serial = time.perf_counter() - t0
    t0 = time.perf_counter()
    s2 = b.train(PRIORS, options=opts)
    t_batched = time.perf_counter() - t0
    assert s1.nit == s2.nit == 12 and s1.nfev == s2.nfev
    np.testing.assert_allclose(s2.x, s1.x, rtol=1e-9)
    print(f"train, N = 4096, 12 iterations ({s1.nfev} evaluations): serial {t_serial * 1e3:.1f} ms, batched {t_batched * 1e3:.1f} ms "
          f"({s2.nbatches} device calls, {s2.nfev_speculative} rows)")
    assert t_batched < 0.75 * t_serial
