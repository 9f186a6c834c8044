"""
Multi-order model: the sum of independent per-order log-likelihoods.

The reference ships only a stub (``class EchelleModel: pass``, Starfish/models/echelle_model.py:1-2;
multi-order fitting "will be added back", docs/conversion.rst:7-8).  Orders are statistically
independent given the stellar parameters (docs/intro.rst:71-73), so lnL = sum over orders; each
order is a :class:`SpectrumModel` whose (walker x order) units are evaluated in batched device passes
and may live on different GPUs (``devices``) with only a host-side sum -- no collective.
"""
import numpy as np

from ..spectrum import Spectrum
from .spectrum_model import SpectrumModel


class EchelleModel:
    def __init__(self, emulator, data, grid_params, devices=None, name="EchelleModel", **params):
        """``params`` are shared by every order (vz, vsini, log_scale, global_cov, cheb, ...);
        per-order overrides can be set afterwards on ``self.orders[i]``."""
        self.name = name
        self.orders = []
        for i, order in enumerate(data):
            single = Spectrum(order._wave, order._flux, order._sigma, order.mask, name=f"{data.name}[{i}]")
            dev = None if devices is None else devices[i % len(devices)]
            kw = {k: (dict(v) if isinstance(v, dict) else ([dict(x) for x in v] if k == "local_cov" else
                       (list(v) if isinstance(v, (list, tuple)) else v))) for k, v in params.items()}
            self.orders.append(SpectrumModel(emulator, single, grid_params, device=dev, name=f"{name}[{i}]", **kw))

    def __len__(self):
        return len(self.orders)

    @property
    def labels(self):
        return self.orders[0].labels

    def freeze(self, names):
        for m in self.orders:
            m.freeze(names)

    def thaw(self, names):
        for m in self.orders:
            m.thaw(names)

    def get_param_vector(self):
        return self.orders[0].get_param_vector()

    def set_param_vector(self, P):
        for m in self.orders:
            m.set_param_vector(P)

    def log_likelihood(self, priors=None):
        """Sum of the per-order likelihoods; the prior is counted once."""
        total = self.orders[0].log_likelihood(priors)
        for m in self.orders[1:]:
            total += m.log_likelihood(None)
        return total

    def log_likelihood_batch(self, P, priors=None):
        """lnL (B,) for B shared parameter vectors: sum over orders of the batched per-order passes."""
        P = np.atleast_2d(np.asarray(P, dtype=np.float64))
        total = self.orders[0].log_likelihood_batch(P, priors)
        for m in self.orders[1:]:
            total = total + m.log_likelihood_batch(P, None)
        return total
