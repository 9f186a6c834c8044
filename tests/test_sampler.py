"""CPU test of the vectorised stretch-move sampler on an analytic target; GPU test of the whole loop."""
import numpy as np
import pytest

from starfish_amd.samplers import EnsembleSampler


def test_sampler_recovers_gaussian_moments():
    mu = np.array([1.0, -2.0, 0.5])
    sig = np.array([0.5, 2.0, 1.0])
    calls = []

    def log_prob(P):
        calls.append(P.shape)
        return -0.5 * (((P - mu) / sig) ** 2).sum(axis=1)

    s = EnsembleSampler(32, 3, log_prob, seed=3)
    p0 = mu + 0.1 * np.random.default_rng(0).standard_normal((32, 3))
    s.run_mcmc(p0, 1500)
    flat = s.get_chain(discard=300, flat=True)
    assert np.all(np.abs(flat.mean(axis=0) - mu) < 0.15 * sig)
    assert np.all(np.abs(flat.std(axis=0) / sig - 1) < 0.15)
    assert 0.2 < s.acceptance_fraction.mean() < 0.9
    assert calls[0] == (32, 3) and calls[1] == (16, 3)  # half-ensembles are evaluated as one batch
    with pytest.raises(ValueError):
        EnsembleSampler(5, 3, log_prob)
    with pytest.raises(ValueError):
        s.run_mcmc(p0[:10], 1)


def test_sampler_never_moves_into_minus_infinity():
    def log_prob(P):
        out = -0.5 * (P**2).sum(axis=1)
        out[P[:, 0] < 0] = -np.inf
        return out

    s = EnsembleSampler(16, 2, log_prob, seed=1)
    p0 = np.abs(np.random.default_rng(1).standard_normal((16, 2))) + 0.1
    s.run_mcmc(p0, 200)
    assert (s.get_chain()[..., 0] >= 0).all()


@pytest.mark.gpu
def test_sampler_drives_the_batched_model():
    import scipy.stats as st

    from starfish_amd import Spectrum, synth
    from starfish_amd.emulator import Emulator
    from starfish_amd.models import SpectrumModel

    o = synth.make_order(N=256, m=4, seed=5)
    emu = Emulator(o["grid_points"], o["param_names"], o["emu_wl"], o["weights"], o["eigenspectra"],
                   o["w_hat"], o["flux_mean"], o["flux_std"], o["factors"])
    emu._trained = True
    c = synth.centre_params(o)
    gp = c.pop("grid_params")
    model = SpectrumModel(emu, Spectrum(o["wave"], o["flux"], sigmas=o["sigma"]), grid_params=gp, **c)
    model.freeze(["local_cov", "cheb", "logg", "Z"])
    priors = {"vsini": st.uniform(0, 200), "T": st.uniform(6000, 200)}
    ndim = len(model.labels)
    nwalkers = 4 * ndim
    p0 = model.get_param_vector() + 1e-3 * np.random.default_rng(0).standard_normal((nwalkers, ndim))
    s = EnsembleSampler(nwalkers, ndim, lambda P: model.log_likelihood_batch(P, priors), seed=0)
    x, lp = s.run_mcmc(p0, 20)
    assert np.isfinite(lp).all() and s.get_chain().shape == (20, nwalkers, ndim)
    # the chain's log-probabilities are exactly what the scalar reference-style path returns
    model.set_param_vector(x[0])
    assert abs(model.log_likelihood(priors) - lp[0]) <= 1e-9 * abs(lp[0])
    assert lp.max() >= s.get_log_prob()[0].max() - 1e-9 or s.acceptance_fraction.mean() > 0
