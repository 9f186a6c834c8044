"""
Nelder-Mead with the candidate points of an iteration evaluated as ONE batch.

``SpectrumModel.train`` (Starfish/models/spectrum_model.py:635-696) drives ``scipy.optimize.minimize(method="Nelder-Mead")``
over the scalar log-likelihood: ~10^3 SERIAL evaluations of a batch of one, each bound by the latency of the panel
chain of one matrix (3.8 ms at N = 4096 for 0.08 of the chip).  But every point an iteration of the simplex method may ask
for is a function of the CURRENT simplex alone: the reflection, the expansion and the two contractions lie on the line
through the worst vertex and the centroid of the others.  So they are evaluated speculatively as one batch of four, the
N + 1 vertices of the initial simplex as one batch, and the N vertices of a shrink step as one batch -- a batch of four
matrices costs the device 1.2 x a batch of one.

The DECISIONS are scipy's, made on exactly the values scipy would have asked for and in its order (accept / expand /
contract / shrink rules, tie handling, termination tests, the `maxfev` cut-off in the middle of an iteration, the
coefficients of `adaptive=True`, a user-supplied `initial_simplex`), so for the same function values the iterates, `nit`,
`nfev`, `status` and the final simplex are the ones `scipy.optimize.minimize` returns (tests/test_neldermead.py holds this
against scipy itself, bit for bit, on analytic functions).  Speculative points that the rules never look at are dropped:
their values -- and their failures (a proposal outside the emulator grid) -- have no effect.

Host-side control logic only: every function value comes from ``fbatch`` (the GPU).
"""
import numpy as np
from scipy.optimize import OptimizeResult

SUPPORTED_OPTIONS = frozenset(("maxiter", "maxfev", "xatol", "fatol", "adaptive", "initial_simplex", "return_all", "disp"))


class _Budget(Exception):
    """The evaluation count reached `maxfev` (scipy stops in the middle of the iteration, keeps what it has)."""


def default_simplex(x0):
    """x0 and, per coordinate, x0 with that coordinate enlarged by 5 % (0.00025 where it is zero)."""
    x0 = np.asarray(x0, dtype=np.float64)
    sim = np.tile(x0, (x0.size + 1, 1))
    for k in range(x0.size):
        sim[k + 1, k] = 1.05 * x0[k] if x0[k] != 0 else 0.00025
    return sim


def minimize_neldermead_batched(fbatch, x0, maxiter=None, maxfev=None, xatol=1e-4, fatol=1e-4, adaptive=False,
                                initial_simplex=None, return_all=False, disp=False, callback=None):
    """``fbatch(X)`` -> the function values of the rows of ``X`` (shape (k, N)); it may raise for a row only through
    the optional second return value: ``fbatch`` may return ``(values, raiser)`` where ``raiser(i)`` raises the error
    of row i or returns -- it is called for the rows whose values the method actually uses, in scipy's order.
    Returns an ``OptimizeResult`` with scipy's fields plus ``nbatches`` (device calls) and ``nfev_speculative`` (rows
    evaluated, used or not) and ``last_x`` (the last point whose value was used: where scipy's objective was last called)."""
    x0 = np.atleast_1d(np.asarray(x0, dtype=np.float64)).ravel()
    if initial_simplex is None:
        sim = default_simplex(x0)
    else:
        sim = np.array(np.atleast_2d(initial_simplex), dtype=np.float64)
        if sim.ndim != 2 or sim.shape[0] != sim.shape[1] + 1:
            raise ValueError("`initial_simplex` should be an array of shape (N+1,N)")
        if x0.size != sim.shape[1]:
            raise ValueError("Size of `initial_simplex` is not consistent with `x0`")
    N = sim.shape[1]
    if adaptive:
        rho, chi, psi, sigma = 1.0, 1.0 + 2.0 / N, 0.75 - 1.0 / (2.0 * N), 1.0 - 1.0 / N
    else:
        rho, chi, psi, sigma = 1.0, 2.0, 0.5, 0.5
    if maxiter is None and maxfev is None:
        maxiter = maxfev = N * 200
    elif maxiter is None:
        maxiter = N * 200 if maxfev == np.inf else np.inf
    elif maxfev is None:
        maxfev = N * 200 if maxiter == np.inf else np.inf

    count = dict(used=0, rows=0, batches=0)
    last_x = [sim[0].copy()]

    def evaluate(X):
        out = fbatch(X)
        vals, raiser = out if isinstance(out, tuple) else (out, None)
        vals = np.asarray(vals, dtype=np.float64)
        if vals.shape != (len(X),):
            raise ValueError("fbatch must return one value per row")
        count["rows"] += len(X)
        count["batches"] += 1

        def use(i):
            # the i-th row's value, as scipy's wrapped objective would hand it out: refused once the budget is spent
            if count["used"] >= maxfev:
                raise _Budget()
            count["used"] += 1
            last_x[0] = np.array(X[i], copy=True)
            if raiser is not None:
                raiser(i)
            return float(vals[i])

        return use

    fsim = np.full(N + 1, np.inf)
    allvecs = [sim[0]] if return_all else None  # (scipy records the first vertex BEFORE the simplex is sorted)
    use = evaluate(sim)  # the N + 1 vertices: ONE batch
    try:
        for k in range(N + 1):
            fsim[k] = use(k)
    except _Budget:
        pass
    order = np.argsort(fsim)  # (scipy sorts twice here; the second sort of a sorted array is the identity up to ties:
    sim, fsim = sim[order], fsim[order]  # argsort's default quicksort is deterministic for equal input)
    order = np.argsort(fsim)
    sim, fsim = sim[order], fsim[order]
    nit = 1
    while count["used"] < maxfev and nit < maxiter:
        halt = False
        try:
            if np.max(np.abs(sim[1:] - sim[0])) <= xatol and np.max(np.abs(fsim[0] - fsim[1:])) <= fatol:
                halt = True  # converged (the bookkeeping below still runs once, as scipy's `finally` does)
            else:
                xbar = np.add.reduce(sim[:-1], 0) / N
                worst = sim[-1]
                cand = np.stack([
                    (1 + rho) * xbar - rho * worst,              # reflection
                    (1 + rho * chi) * xbar - rho * chi * worst,  # expansion
                    (1 + psi * rho) * xbar - psi * rho * worst,  # outside contraction
                    (1 - psi) * xbar + psi * worst,              # inside contraction
                ])
                use = evaluate(cand)  # all four: ONE batch; the rules below look at one or two of them
                fxr = use(0)
                shrink = False
                if fxr < fsim[0]:
                    fxe = use(1)
                    if fxe < fxr:
                        sim[-1], fsim[-1] = cand[1], fxe
                    else:
                        sim[-1], fsim[-1] = cand[0], fxr
                elif fxr < fsim[-2]:
                    sim[-1], fsim[-1] = cand[0], fxr
                elif fxr < fsim[-1]:
                    fxc = use(2)
                    if fxc <= fxr:
                        sim[-1], fsim[-1] = cand[2], fxc
                    else:
                        shrink = True
                else:
                    fxcc = use(3)
                    if fxcc < fsim[-1]:
                        sim[-1], fsim[-1] = cand[3], fxcc
                    else:
                        shrink = True
                if shrink:
                    # (scipy moves and evaluates vertex by vertex: when the budget runs out half-way, the vertex whose
                    # evaluation was refused has moved and keeps its old value, the later ones have not moved)
                    new = sim[0] + sigma * (sim[1:] - sim[0])
                    use = None
                    for j in range(1, N + 1):
                        sim[j] = new[j - 1]
                        if use is None:
                            if count["used"] >= maxfev:
                                raise _Budget()
                            use = evaluate(new)  # the N shrunk vertices: ONE batch
                        fsim[j] = use(j - 1)
                nit += 1
        except _Budget:
            pass
        order = np.argsort(fsim)
        sim, fsim = sim[order], fsim[order]
        if return_all:
            allvecs.append(sim[0])
        if callback is not None:
            try:
                if _wants_result(callback):
                    callback(OptimizeResult(x=sim[0], fun=fsim[0]))
                else:
                    callback(np.copy(sim[0]))
            except StopIteration:
                halt = True
        if halt:
            break

    if count["used"] >= maxfev:
        status, msg = 1, "Maximum number of function evaluations has been exceeded."
    elif nit >= maxiter:
        status, msg = 2, "Maximum number of iterations has been exceeded."
    else:
        status, msg = 0, "Optimization terminated successfully."
    if disp:
        print(msg)
        print(f"         Current function value: {np.min(fsim):f}")
        print(f"         Iterations: {nit}")
        print(f"         Function evaluations: {count['used']} ({count['rows']} rows in {count['batches']} batches)")
    res = OptimizeResult(fun=np.min(fsim), nit=nit, nfev=count["used"], status=status, success=status == 0, message=msg,
                         x=sim[0], final_simplex=(sim, fsim), nbatches=count["batches"], nfev_speculative=count["rows"],
                         last_x=last_x[0])
    if return_all:
        res["allvecs"] = allvecs
    return res


def _wants_result(callback):
    """scipy hands `intermediate_result` to callbacks whose only parameter has that name, the current x otherwise."""
    import inspect

    try:
        params = list(inspect.signature(callback).parameters)
    except (TypeError, ValueError):
        return False
    return params == ["intermediate_result"]


def split_minimize_kwargs(kwargs):
    """The keyword arguments ``SpectrumModel.train`` forwards to ``scipy.optimize.minimize``: (options for the batched
    method, None) when they describe a plain Nelder-Mead run that it covers, (None, reason) otherwise."""
    kw = dict(kwargs)
    method = kw.pop("method", "Nelder-Mead")
    if not isinstance(method, str) or method.lower() != "nelder-mead":
        return None, f"method {method!r}"
    opts = dict(kw.pop("options", None) or {})
    tol = kw.pop("tol", None)
    callback = kw.pop("callback", None)
    if kw.pop("bounds", None) is not None or kw.pop("constraints", ()) not in ((), None):
        return None, "bounds / constraints"
    kw.pop("args", None)
    if kw:
        return None, f"arguments {sorted(kw)}"
    if set(opts) - SUPPORTED_OPTIONS:
        return None, f"options {sorted(set(opts) - SUPPORTED_OPTIONS)}"
    if tol is not None:  # (scipy.optimize.minimize: tol sets both tolerances of Nelder-Mead unless given)
        opts.setdefault("xatol", tol)
        opts.setdefault("fatol", tol)
    opts["callback"] = callback
    return opts, None
