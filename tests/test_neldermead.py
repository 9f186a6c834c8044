"""The batched Nelder-Mead behind SpectrumModel.train (starfish_amd/_neldermead.py) against scipy.optimize.minimize itself:
for the same function values it must make scipy's decisions -- identical iterates, counts, status and final simplex, bit for
bit -- while asking for its points in batches (four candidates per iteration, the initial simplex and every shrink step as
one batch each).  CPU only (analytic objectives); the GPU counterpart on a real model is tests/test_gpu_train.py."""
import numpy as np
import pytest
from scipy.optimize import minimize

from starfish_amd._neldermead import default_simplex, minimize_neldermead_batched, split_minimize_kwargs


def rosen(x):
    x = np.asarray(x)
    return float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2))


def bumpy(x):  # many contractions and shrinks: a narrow curved valley with a ripple
    x = np.asarray(x)
    return float(np.sum(np.abs(x) ** 1.5) + 0.3 * np.sum(np.sin(5 * x) ** 2) + 10 * (x[0] * x[-1] - 0.2) ** 2)


def plateau(x):  # ties: a function that is constant on cells (exercises the <, <= of the rules)
    return float(np.sum(np.floor(np.abs(np.asarray(x)) * 3)))


def noisy(x):  # a rough surface: contractions fail, the simplex SHRINKS (two shrink steps from the start below)
    x = np.asarray(x)
    return float(np.sum(x**2) + 3 * np.abs(np.sin(37 * np.sum(x) + 11 * x[0] * x[-1])))


def with_inf(x):  # +inf outside a box (a walker outside the prior / emulator grid)
    x = np.asarray(x)
    return float("inf") if np.any(np.abs(x) > 3) else rosen(x)


def batched(f):
    calls = []

    def fb(X):
        calls.append(len(X))
        return np.array([f(x) for x in X])

    return fb, calls


CASES = [
    (rosen, np.array([-1.2, 1.0]), {}),
    (rosen, np.array([-1.2, 1.0, 0.7, -0.3, 2.0]), {}),
    (rosen, np.array([-1.2, 1.0, 0.7, -0.3, 2.0]), {"adaptive": True}),
    (rosen, np.zeros(4), {"maxiter": 37}),
    (rosen, np.array([0.3, -0.4, 1.5]), {"maxfev": 61}),
    (rosen, np.array([0.3, -0.4, 1.5]), {"maxfev": 2}),          # the budget ends inside the initial simplex
    (bumpy, np.array([1.0, -2.0, 0.5, 0.0, 1.5, -0.7]), {"xatol": 1e-7, "fatol": 1e-7}),
    (bumpy, np.array([1.0, -2.0, 0.5]), {"maxfev": 45, "xatol": 1e-9, "fatol": 1e-9}),
    (plateau, np.array([1.3, -2.2, 0.7]), {}),
    (with_inf, np.array([2.5, -2.8, 2.9]), {}),
    (noisy, np.array([1.0, -2.0, 0.5]), {}),
    (noisy, np.array([1.0, -2.0, 0.5]), {"maxfev": 100}),
    (rosen, np.array([5.0, 5.0]), {"initial_simplex": np.array([[5.0, 5.0], [4.0, 5.5], [5.5, 4.0]])}),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_same_iterates_counts_and_status_as_scipy(case):
    f, x0, opts = CASES[case]
    want = minimize(f, x0, method="Nelder-Mead", options=dict(opts, return_all=True))
    fb, calls = batched(f)
    got = minimize_neldermead_batched(fb, x0, return_all=True, **opts)
    assert (got.nit, got.nfev, got.status, got.success) == (want.nit, want.nfev, want.status, want.success)
    assert got.message == want.message
    np.testing.assert_array_equal(got.x, want.x)
    assert got.fun == want.fun
    np.testing.assert_array_equal(got.final_simplex[0], want.final_simplex[0])
    np.testing.assert_array_equal(got.final_simplex[1], want.final_simplex[1])
    assert len(got.allvecs) == len(want.allvecs)
    for a, b in zip(got.allvecs, want.allvecs):
        np.testing.assert_array_equal(a, b)
    # the batching: N + 1 vertices first, then four candidates per iteration, N per shrink
    N = len(x0)
    assert calls[0] == N + 1 and set(calls[1:]) <= {4, N}
    assert got.nbatches == len(calls) and got.nfev_speculative == sum(calls) >= got.nfev
    assert got.nbatches <= got.nit + 1 + calls.count(N) + 1


def test_shrink_steps_occur_in_the_cases_above_and_are_one_batch():
    fb, calls = batched(noisy)
    minimize_neldermead_batched(fb, np.array([1.0, -2.0, 0.5]))
    assert calls[1:].count(3) == 2


def test_errors_of_unused_speculative_points_are_ignored_and_those_of_used_points_raise():
    """fbatch may return (values, raiser): raiser(i) is called only for rows whose value the rules use."""
    asked = []

    def fb(X):
        vals = np.array([rosen(x) for x in X])

        def raiser(i):
            asked.append(tuple(X[i]))
            if X[i][0] > 50:
                raise ValueError("outside the grid")

        return vals, raiser

    res = minimize_neldermead_batched(fb, np.array([-1.2, 1.0]), maxiter=30)
    assert len(asked) == res.nfev < res.nfev_speculative
    np.testing.assert_array_equal(res.last_x, asked[-1])
    # a start whose first reflection is used and invalid raises like the scalar objective would
    with pytest.raises(ValueError, match="outside the grid"):
        minimize_neldermead_batched(fb, np.array([60.0, 1.0]), maxiter=5)


def test_callback_conventions_and_stop_iteration():
    seen = []

    def cb(xk):
        seen.append(np.array(xk))

    want_seen = []
    minimize(rosen, [-1.2, 1.0], method="Nelder-Mead", callback=lambda xk: want_seen.append(np.array(xk)), options=dict(maxiter=12))
    fb, _ = batched(rosen)
    minimize_neldermead_batched(fb, [-1.2, 1.0], maxiter=12, callback=cb)
    assert len(seen) == len(want_seen)
    for a, b in zip(seen, want_seen):
        np.testing.assert_array_equal(a, b)

    def halting(intermediate_result):
        if intermediate_result.fun < 1.0:
            raise StopIteration

    want = minimize(rosen, [-1.2, 1.0], method="Nelder-Mead", callback=halting)
    got = minimize_neldermead_batched(fb, [-1.2, 1.0], callback=halting)
    assert got.nit == want.nit and got.nfev == want.nfev
    np.testing.assert_array_equal(got.x, want.x)


def test_default_simplex_and_kwarg_split():
    np.testing.assert_array_equal(default_simplex([2.0, 0.0]), [[2.0, 0.0], [2.1, 0.0], [2.0, 0.00025]])
    opts, why = split_minimize_kwargs({"options": {"maxiter": 10}, "tol": 1e-3})
    assert why is None and opts["maxiter"] == 10 and opts["xatol"] == opts["fatol"] == 1e-3 and opts["callback"] is None
    assert split_minimize_kwargs({"method": "BFGS"})[0] is None
    assert split_minimize_kwargs({"bounds": [(0, 1)]})[0] is None
    assert split_minimize_kwargs({"options": {"maxiter": 3, "unknown_knob": 1}})[0] is None
    assert split_minimize_kwargs({"method": "nelder-mead"})[1] is None
    with pytest.raises(ValueError):
        minimize_neldermead_batched(lambda X: np.zeros(len(X)), [0.0, 0.0], initial_simplex=np.zeros((2, 2)))
