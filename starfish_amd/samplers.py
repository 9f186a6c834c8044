"""
A small affine-invariant ensemble sampler (Goodman & Weare 2010 stretch move, the algorithm behind
``emcee.EnsembleSampler``) that feeds whole half-ensembles to a VECTORISED log-probability function,
i.e. ``SpectrumModel.log_likelihood_batch`` -- the caller contract of the reference's driver loop
(examples/single.ipynb:436-466, 528-546).  ``emcee`` itself is not installed on either box; with it
installed, ``emcee.EnsembleSampler(..., vectorize=True)`` accepts the same function unchanged.

Host-side control logic only: every likelihood evaluation happens on the GPU inside ``log_prob_fn``.
"""
import numpy as np


class EnsembleSampler:
    """emcee-like surface: ``EnsembleSampler(nwalkers, ndim, log_prob_fn, a=2.0, seed=None)``,
    ``run_mcmc(p0, nsteps)``, ``get_chain(discard, flat)``, ``get_log_prob``, ``acceptance_fraction``.
    ``log_prob_fn`` receives an (n, ndim) block and returns n log-probabilities (-inf allowed)."""

    def __init__(self, nwalkers, ndim, log_prob_fn, a=2.0, seed=None):
        if nwalkers % 2 or nwalkers < 2 * ndim:
            raise ValueError("need an even number of walkers, at least 2 * ndim")
        self.nwalkers, self.ndim, self.a = nwalkers, ndim, float(a)
        self.log_prob_fn = log_prob_fn
        self.rng = np.random.default_rng(seed)
        self.chain = np.empty((0, nwalkers, ndim))
        self.log_prob = np.empty((0, nwalkers))
        self.naccepted = np.zeros(nwalkers)
        self.iteration = 0

    def _half_step(self, x, lp, active, other):
        ns = len(active)
        # z ~ g(z) proportional to 1/sqrt(z) on [1/a, a]
        z = ((self.a - 1.0) * self.rng.random(ns) + 1.0) ** 2 / self.a
        partner = other[self.rng.integers(len(other), size=ns)]
        prop = x[partner] + z[:, None] * (x[active] - x[partner])
        lp_new = np.asarray(self.log_prob_fn(prop), dtype=np.float64)
        if np.any(np.isnan(lp_new)):
            raise ValueError("log_prob_fn returned NaN")
        log_ratio = (self.ndim - 1.0) * np.log(z) + lp_new - lp[active]
        accept = np.log(self.rng.random(ns)) < log_ratio
        idx = active[accept]
        x[idx] = prop[accept]
        lp[idx] = lp_new[accept]
        self.naccepted[idx] += 1

    def run_mcmc(self, p0, nsteps):
        x = np.array(p0, dtype=np.float64)
        if x.shape != (self.nwalkers, self.ndim):
            raise ValueError("p0 must have shape (nwalkers, ndim)")
        lp = np.asarray(self.log_prob_fn(x), dtype=np.float64)
        if not np.all(np.isfinite(lp)):
            raise ValueError("initial state has non-finite log-probability")
        chain = np.empty((nsteps, self.nwalkers, self.ndim))
        lps = np.empty((nsteps, self.nwalkers))
        first = np.arange(self.nwalkers // 2)
        second = np.arange(self.nwalkers // 2, self.nwalkers)
        for t in range(nsteps):
            self._half_step(x, lp, first, second)
            self._half_step(x, lp, second, first)
            chain[t], lps[t] = x, lp
        self.chain = np.concatenate([self.chain, chain])
        self.log_prob = np.concatenate([self.log_prob, lps])
        self.iteration += nsteps
        return x, lp

    @property
    def acceptance_fraction(self):
        return self.naccepted / max(1, self.iteration)

    def get_chain(self, discard=0, flat=False):
        c = self.chain[discard:]
        return c.reshape(-1, self.ndim) if flat else c

    def get_log_prob(self, discard=0, flat=False):
        lp = self.log_prob[discard:]
        return lp.reshape(-1) if flat else lp
