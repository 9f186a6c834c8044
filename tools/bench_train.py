"""SpectrumModel.train at N = 4096: the reference's serial Nelder-Mead loop (scipy over the scalar likelihood: B = 1 device
launches) against the batched simplex (starfish_amd/_neldermead.py: the four candidates of an iteration as one device
batch).  Same iterations, same decisions.
    python tools/bench_train.py [N] [iterations]"""
import sys
import time

import numpy as np
import scipy.stats as st

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from starfish_amd import synth  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    o = synth.make_order(N=N)
    a, b = synth.build_model(o), synth.build_model(o)
    for m in (a, b):
        m.log_likelihood()
        m.log_likelihood_batch(np.tile(m.get_param_vector(), (14, 1)))
        m.log_likelihood_batch(np.tile(m.get_param_vector(), (4, 1)))
    # (scipy's default simplex enlarges T by 5 %: outside the emulator grid, where the objective raises -- in the reference
    # too; the prior keeps such vertices at -inf)
    priors = {"T": st.uniform(6000, 200), "vsini": st.uniform(0, 500)}
    rows = []
    for label, model, kw in (("serial (scipy, B = 1 per evaluation)", a, dict(batch_simplex=False)), ("batched simplex", b, {})):
        for n_it in (1, iters):  # (1 iteration = the initial simplex alone: N + 1 evaluations)
            x0 = model.get_param_vector().copy()
            t0 = time.perf_counter()
            s = model.train(priors, options=dict(maxiter=n_it), **kw)
            dt = time.perf_counter() - t0
            model.set_param_vector(x0)
            rows.append((label, n_it, s.nit, s.nfev, getattr(s, "nbatches", s.nfev), dt * 1e3))
    print(f"SpectrumModel.train, N = {N}, 13 thawed parameters (ms wall clock, host logic included)")
    for label, n_it, nit, nfev, calls, ms in rows:
        print(f"  {label:40s} maxiter {n_it:4d}: nit {nit:4d} nfev {nfev:4d} device calls {calls:4d}  {ms:9.1f} ms")
    (s1, b1), (sN, bN) = (rows[0][5], rows[2][5]), (rows[1][5], rows[3][5])
    print(f"  initial simplex (14 evaluations): serial {s1:.1f} ms, batched {b1:.1f} ms  -> {s1 / b1:.2f} x")
    per_s, per_b = (sN - s1) / max(1, rows[1][2] - 1), (bN - b1) / max(1, rows[3][2] - 1)
    print(f"  per iteration after it: serial {per_s:.2f} ms ({(rows[1][3] - 14) / max(1, rows[1][2] - 1):.2f} evaluations), batched {per_b:.2f} ms -> {per_s / per_b:.2f} x")
    print(f"  whole run of {iters} iterations: {sN / bN:.2f} x")


if __name__ == "__main__":
    main()
