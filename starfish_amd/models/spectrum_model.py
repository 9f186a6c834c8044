"""
``SpectrumModel`` -- the single-order forward model and Gaussian log-likelihood, with the reference's
Python surface (reference: Starfish/models/spectrum_model.py) so an existing emcee / scipy loop runs
unchanged, plus ``log_likelihood_batch`` which evaluates a whole block of walkers in one launch
sequence on the MI355X.

All numerics (transform chain, emulator query, fused covariance fill, batched Cholesky, solve) run in
the HIP kernels behind ``include/starfish_amd.h``; this file only keeps the parameter book-keeping
(FlatterDict store, labels, freeze / thaw, caches) and packs parameter rows for the C-ABI.
"""
import logging
import zlib
from collections import deque

import numpy as np

from .. import _device as D
from .._flatdict import FlatterDict
from ..emulator import Emulator
from ..spectrum import Spectrum
from ..transforms import resample
from ..utils import calculate_dv, create_log_lam_grid


try:  # content hash of the observation arrays (checked on every evaluation): xxh3 is ~20x faster than crc32
    from xxhash import xxh3_64_intdigest as _hash64
except ImportError:  # pragma: no cover
    _hash64 = zlib.crc32


def _digest(a):
    return _hash64(np.ascontiguousarray(a))


class SpectrumModel:
    """
    A single-order spectrum model (Starfish/models/spectrum_model.py:26-181).

    Parameters
    ----------
    emulator : Emulator or str
    data : Spectrum or str
        exactly one order (``ValueError`` otherwise, spectrum_model.py:141-144)
    grid_params : array-like
        values for ``emulator.param_names``
    max_deque_len, norm, name : as in the reference
    device : int, optional
        HIP device index holding this order's buffers (default: current device)
    **params :
        ``vz, vsini, Av, Rv, log_scale, global_cov={log_amp, log_ls},
        local_cov=[{mu, log_amp, log_sigma}, ...], cheb=[c1, c2, ...]``
    """

    _PARAMS = ["vz", "vsini", "Av", "Rv", "log_scale", "global_cov", "local_cov", "cheb"]
    _GLOBAL_PARAMS = ["log_amp", "log_ls"]
    _LOCAL_PARAMS = ["mu", "log_amp", "log_sigma"]

    def __init__(self, emulator, data, grid_params, max_deque_len=100, norm=False, name="SpectrumModel",
                 device=None, solver="dense", emulator_cov="code", **params):
        if isinstance(emulator, str):
            emulator = Emulator.load(emulator)
        if isinstance(data, str):
            data = Spectrum.load(data)
        if len(data) > 1:
            raise ValueError("Multiple orders detected in data, please use EchelleModel")

        self.emulator = emulator
        self.data_name = data.name
        self.data = data[0]
        self._device_index = device
        #: "dense" = the reference's algorithm (batched N x N Cholesky); "auto" = band + rank-m Woodbury
        #: solve whenever the covariance support fits the device window (same value to rounding, ~20x
        #: faster at N = 4096), dense otherwise; "banded" = structured solve only (info -4 if too wide)
        if solver not in ("dense", "banded", "auto"):
            raise ValueError("solver must be 'dense', 'banded' or 'auto'")
        self.solver = solver
        #: form of the emulator term of the covariance.  "code" (default, the parity target): X^T Sigma_w^-1 X, what the
        #: reference COMPUTES (spectrum_model.py:334-335: cho_solve of the weight covariance); "paper": X^T Sigma_w X, the
        #: form printed in the reference's docs (docs/api/emulator.rst:105) and in Czekala et al. (2015).
        if emulator_cov not in ("code", "paper"):
            raise ValueError("emulator_cov must be 'code' or 'paper'")
        self.emulator_cov = emulator_cov

        dv = calculate_dv(self.data.wave)
        self.min_dv_wave = create_log_lam_grid(dv, self.emulator.wl.min(), self.emulator.wl.max())["wl"]
        self._bulk_fluxes = None  # resampled lazily on the device (spectrum_model.py:154-156)
        self._dev = None
        self._dev_v11 = None
        self._dev_data = None

        self.residuals = deque(maxlen=max_deque_len)

        # cheb is stored as {"1": c1, "2": c2, ...}; re-inserting moves it behind the other kwargs
        if "cheb" in params:
            chebs = params.pop("cheb")
            params["cheb"] = {str(i): c for i, c in enumerate(chebs, start=1)}
        self.params = FlatterDict(params)
        self.frozen = []
        self.name = name
        self.norm = norm

        self.n_grid_params = len(grid_params)
        self.grid_params = grid_params

        self._lnprob = None
        self._glob_snapshot = None  # hyper-parameters the cached global kernel was built with
        self._loc_snapshot = None
        self._log_scale = params.get("log_scale", None)
        self.last_info = None

        self.log = logging.getLogger(self.__class__.__name__)

    # ------------------------------------------------------------------ lazily built device state
    @property
    def bulk_fluxes(self):
        """Emulator bulk fluxes resampled onto ``min_dv_wave`` (spectrum_model.py:154-156)."""
        if self._bulk_fluxes is None:
            self._bulk_fluxes = resample(self.emulator.wl, self.emulator.bulk_fluxes, self.min_dv_wave)
        return self._bulk_fluxes

    def _device(self):
        emu = self.emulator
        # the observation lives in HBM; the reference reads self.data on every evaluation
        # (spectrum_model.py:365-377), so a replaced or edited flux / sigma must reach the device copy
        d = self.data
        key = tuple(_digest(a) for a in (d.wave, d.flux, d.sigma))
        stale = self._dev is None or self._dev_v11 is not emu.v11 or key != self._dev_data
        if stale:
            if self._dev_data is not None and key[0] != self._dev_data[0]:
                # a new wavelength grid: what the constructor derived from it (spectrum_model.py:150-156) follows
                dv = calculate_dv(d.wave)
                self.min_dv_wave = create_log_lam_grid(dv, emu.wl.min(), emu.wl.max())["wl"]
                self._bulk_fluxes = None
            self._dev_data = key
            self._dev = D.DeviceOrder(
                self.data.wave, self.data.flux, self.data.sigma, self.min_dv_wave, self.bulk_fluxes,
                emu.grid_points, emu.variances, emu.lengthscales, emu.v11, emu.w_hat,
                device=self._device_index, emu_factor=emu.v11_factor(),
            )
            self._dev_v11 = emu.v11
        return self._dev

    # ------------------------------------------------------------------ covariance caches
    # The reference caches the N x N kernel matrices while their group is frozen
    # (spectrum_model.py:341-363).  Here the kernels are re-evaluated inside the fused fill, so the
    # cache keeps the hyper-parameters the matrix WOULD have been built with -- same observable
    # behaviour (a frozen group keeps using the values seen when the cache was filled).
    @property
    def _glob_cov(self):
        if self._glob_snapshot is None:
            return None
        from .kernels import global_covariance_matrix

        la, ll = self._glob_snapshot
        return global_covariance_matrix(self.data.wave, np.exp(la), np.exp(ll))

    @_glob_cov.setter
    def _glob_cov(self, value):
        if value is not None:
            raise AttributeError("_glob_cov can only be reset to None")
        self._glob_snapshot = None

    @property
    def _loc_cov(self):
        if self._loc_snapshot is None:
            return None
        from .kernels import local_covariance_matrix

        out = 0
        for mu, la, ls in self._loc_snapshot:
            out = out + local_covariance_matrix(self.data.wave, np.exp(la), mu, np.exp(ls))
        return out

    @_loc_cov.setter
    def _loc_cov(self, value):
        if value is not None:
            raise AttributeError("_loc_cov can only be reset to None")
        self._loc_snapshot = None

    # ------------------------------------------------------------------ parameter views
    @property
    def grid_params(self):
        """numpy.ndarray : emulator parameters in ``emulator.param_names`` order."""
        return np.array([self.params[key] for key in self.emulator.param_names])

    @grid_params.setter
    def grid_params(self, values):
        for key, value in zip(self.emulator.param_names, values):
            if key not in self.frozen:
                self.params[key] = value

    @property
    def cheb(self):
        """numpy.ndarray : c1, c2, ... (c0 is fixed to 1)."""
        return np.array(self.params["cheb"].values())

    @cheb.setter
    def cheb(self, values):
        if "cheb" in self.frozen:
            return
        for key, value in zip(self.params["cheb"], values):
            if key not in self.frozen:
                self.params["cheb"][key] = value

    @property
    def labels(self):
        """tuple of str : thawed parameter names, defining the meaning of a parameter vector."""
        return tuple(self.get_param_dict(flat=True).keys())

    def __getitem__(self, key):
        if key == "cheb":
            return list(self.params[key].values())
        return self.params[key]

    def __setitem__(self, key, value):
        if ":" in key:
            group, rest = key.split(":", 1)
            leaf = rest.split(":")[-1]
            if group == "global_cov" and leaf in self._GLOBAL_PARAMS:
                self.params[key] = value
            elif group == "local_cov" and leaf in self._LOCAL_PARAMS:
                self.params[key] = value
            elif group == "cheb":
                if "cheb" in self.params:
                    idx = int(rest)
                    if idx == 0:
                        raise KeyError("cannot change constant Chebyshev term")
                    for i in range(len(self.params[group]) + 1, idx + 1):
                        self.params[f"{group}:{i}"] = 0  # widen with zeros (spectrum_model.py:238-247)
                self.params[key] = value
            else:
                raise KeyError(f"{key} not recognized")
        elif key == "cheb":
            self.params[key] = {str(i): c for i, c in enumerate(value, start=1)}
        elif key in [*self._PARAMS, *self.emulator.param_names]:
            self.params[key] = value
        else:
            raise KeyError(f"{key} not recognized")

    def __delitem__(self, key):
        if key not in self.params:
            raise KeyError(f"{key} not in params")
        if key == "global_cov":
            self._glob_snapshot = None
            self.frozen = [k for k in self.frozen if not k.startswith("global_cov")]
        elif key == "local_cov":
            self._loc_snapshot = None
            self.frozen = [k for k in self.frozen if not k.startswith("local_cov")]
        del self.params[key]
        if key in self.frozen:
            self.frozen.remove(key)

    def get_param_dict(self, flat=False):
        """Thawed parameters, nested or flat (``'local_cov:0:mu'``)."""
        out = FlatterDict()
        for key, val in self.params.items():
            if key not in self.frozen:
                out[key] = val
        return out if flat else out.as_dict()

    def set_param_dict(self, params):
        """Update (never add) parameters; frozen keys are left untouched."""
        for key, val in FlatterDict(params).items():
            if key not in self.frozen:
                self.params[key] = val

    def get_param_vector(self):
        return np.array(list(self.get_param_dict(flat=True).values()))

    def set_param_vector(self, params):
        labels = self.labels
        if len(params) != len(labels):
            raise ValueError("Param Vector does not match length of thawed parameters")
        self.set_param_dict(dict(zip(labels, params)))

    # parameter groups: name -> attribute holding the hyper-parameter snapshot that freezing resets
    _GROUPS = {"global_cov": "_glob_snapshot", "local_cov": "_loc_snapshot", "cheb": None}

    def _group_members(self, group):
        """Flat keys stored below a group, in storage order ('local_cov:0:mu', 'cheb:2', ...)."""
        prefix = group + ":"
        return [key for key in self.params.keys() if key.startswith(prefix)]

    def freeze(self, names):
        """Remove parameters from the sampled vector; they keep their value (spectrum_model.py:495-549).
        A group name freezes all of its members and forgets the group's cached covariance; ``"all"``
        freezes everything that is thawed and keeps the caches."""
        names = [str(n) for n in np.atleast_1d(names)]
        if names[0] == "all":
            self.frozen += [key for key in self.labels if key not in self.frozen]
            self.frozen += [group for group in self._GROUPS if group in self.params]
            return
        for name in names:
            if name in self._GROUPS:
                self.frozen.append(name)
                if self._GROUPS[name]:
                    setattr(self, self._GROUPS[name], None)
                self.frozen += [key for key in self._group_members(name) if key not in self.frozen]
            elif name not in self.frozen and name in self.params:
                self.frozen.append(name)

    def thaw(self, names):
        """Opposite of :meth:`freeze` (spectrum_model.py:551-590); thawing a group that is not frozen
        raises ``ValueError`` like ``list.remove``."""
        names = [str(n) for n in np.atleast_1d(names)]
        if names[0] == "all":
            self.frozen = []
            return
        for name in names:
            if name in self._GROUPS:
                for key in [name, *self._group_members(name)]:
                    self.frozen.remove(key)
            elif name in self.frozen:
                self.frozen.remove(name)

    def _local_kernels(self):
        loc = self.params.as_dict().get("local_cov", [])
        return list(loc.values()) if isinstance(loc, dict) else list(loc)

    # ------------------------------------------------------------------ packing for the C-ABI
    def _model_desc(self, dev):
        n_local = len(self._local_kernels()) if "local_cov" in self.params else 0
        n_cheb = len(self.params["cheb"]) if "cheb" in self.params else 0
        return dev.model_desc(
            "vsini" in self.params, "vz" in self.params, "log_scale" in self.params,
            "global_cov" in self.params, n_local, n_cheb, use_sigma_w=self.emulator_cov == "paper",
            has_av="Av" in self.params,
        )

    def _slot_of(self, dev, md):
        """flat parameter key -> column of the C-ABI parameter row (None: not used on the device)."""
        P = dev.P
        slots = {"vsini": 0, "vz": 1, "log_scale": 2, "global_cov:log_amp": 4, "global_cov:log_ls": 5}
        for i, key in enumerate(self.emulator.param_names):
            slots[key] = 6 + i
        if "cheb" in self.params:
            for pos, key in enumerate(self.params["cheb"].keys()):  # positional, as [1, *self.cheb]
                slots[f"cheb:{key}"] = 6 + P + pos
        off_local = 6 + P + md.n_cheb
        for i in range(md.n_local):
            for j, leaf in enumerate(self._LOCAL_PARAMS):
                slots[f"local_cov:{i}:{leaf}"] = off_local + 3 * i + j
        if md.has_av:
            slots["Av"] = off_local + 3 * md.n_local  # Rv is accepted but never passed on (as the reference)
        return slots

    def _pack(self, P=None, update_caches=True):
        """Rows for the C-ABI.  ``P`` is None (current state, one row) or (B, len(labels))."""
        dev = self._device()
        md = self._model_desc(dev)
        slots = self._slot_of(dev, md)
        stride = dev.param_stride(md)
        base = np.zeros(stride)
        base[3] = 1.0
        for key, val in self.params.items():
            s = slots.get(key)
            if s is not None:
                base[s] = val
        if P is None:
            rows = base[None, :].copy()
        else:
            P = np.atleast_2d(np.asarray(P, dtype=np.float64))
            labels = self.labels
            if P.shape[1] != len(labels):
                raise ValueError("Param Vector does not match length of thawed parameters")
            rows = np.tile(base, (P.shape[0], 1))
            for col, key in enumerate(labels):
                s = slots.get(key)
                if s is not None:
                    rows[:, s] = P[:, col]
        if self.norm:  # spectrum_model.py:316-319
            g = rows[:, 6 : 6 + dev.P]
            rows[:, 3] = np.atleast_1d(self.emulator.norm_factor(g)).astype(np.float64)
        # frozen covariance groups keep the hyper-parameters seen when their cache was filled
        if md.has_global:
            if "global_cov" in self.frozen and self._glob_snapshot is not None:
                rows[:, 4:6] = self._glob_snapshot
            elif update_caches:
                self._glob_snapshot = tuple(rows[-1, 4:6])
        if md.n_local:
            lo = 6 + dev.P + md.n_cheb
            if "local_cov" in self.frozen and self._loc_snapshot is not None:
                rows[:, lo : lo + 3 * md.n_local] = np.asarray(self._loc_snapshot).reshape(-1)
            elif update_caches:
                self._loc_snapshot = [tuple(r) for r in rows[-1, lo : lo + 3 * md.n_local].reshape(-1, 3)]
        return dev, md, rows

    @staticmethod
    def _raise_for_info(code):
        code = int(code)
        if code > 0:
            raise np.linalg.LinAlgError(
                f"{code}-th leading minor of the array is not positive definite"
            )
        if code == -3:
            raise np.linalg.LinAlgError(D.INFO_MESSAGES[-3])
        if code == D.INFO_INTERNAL:  # (survived the retry of DeviceOrder.loglike: not a property of the parameters)
            raise RuntimeError(D.INFO_MESSAGES[D.INFO_INTERNAL])
        if code < 0:
            raise ValueError(D.INFO_MESSAGES.get(code, f"device status {code}"))

    # ------------------------------------------------------------------ evaluation
    def __call__(self):
        """(flux, cov) of the current state (spectrum_model.py:277-365); ``cov`` is a fresh N x N array."""
        dev, md, rows = self._pack()
        out = dev.forward(md, rows)
        self._raise_for_info(out["info"][0])
        self._log_scale = float(out["log_scale"][0])
        return out["flux"][0], out["cov"][0]

    def _prior(self, priors):
        lp = 0
        if priors is not None:
            for key, prior in priors.items():
                if key in self.params:
                    lp += prior.logpdf(self[key])
        return lp

    def log_likelihood(self, priors=None):
        """-(logdet + R^T C^-1 R)/2 + log-prior (spectrum_model.py:367-407)."""
        prior_lp = self._prior(priors)
        if not np.isfinite(prior_lp):
            return -np.inf
        dev, md, rows = self._pack()
        out = dev.loglike(md, rows, want_resid=True, solver=self.solver)
        self.last_info = out["info"]
        self._raise_for_info(out["info"][0])
        self._log_scale = float(out["log_scale"][0])
        self.residuals.append(out["resid"][0])
        self._lnprob = float(out["lnl"][0])
        return self._lnprob + prior_lp

    def _batch_prior(self, P, priors):
        """Log-prior of every row of ``P`` (labels order); parameters not in ``labels`` use the current value."""
        labels = self.labels
        prior_lp = np.zeros(P.shape[0])
        if priors:
            current = {k: self[k] for k in priors if k in self.params}
            for key, prior in priors.items():
                if key not in self.params:
                    continue
                if key in labels:
                    prior_lp += np.asarray(prior.logpdf(P[:, labels.index(key)]), dtype=np.float64)
                else:
                    prior_lp += prior.logpdf(current[key])
        return prior_lp

    def log_likelihood_batch(self, P, priors=None, return_info=False):
        """Log-posterior of B parameter vectors (rows of ``P`` in :attr:`labels` order) in one batched
        device pass.  Walkers that fail (outside the emulator grid, vsini <= 0, non-positive-definite
        covariance, non-finite prior) get ``-inf`` instead of raising; ``info`` carries the codes of
        include/starfish_amd.h.  The model's own parameter state is not modified."""
        P = np.atleast_2d(np.asarray(P, dtype=np.float64))
        prior_lp = self._batch_prior(P, priors)
        finite = np.isfinite(prior_lp)
        lnl = np.full(P.shape[0], -np.inf)
        info = np.zeros(P.shape[0], dtype=np.int32)
        if finite.any():
            # (update_caches=False: a batch must not overwrite the hyper-parameter snapshots that a later
            # freeze("all") + log_likelihood() of the model's OWN state relies on)
            dev, md, rows = self._pack(P[finite], update_caches=False)
            out = dev.loglike(md, rows, solver=self.solver)
            lnl[finite] = out["lnl"] + prior_lp[finite]
            info[finite] = out["info"]
        self.last_info = info
        return (lnl, info) if return_info else lnl

    def train(self, priors=None, batch_simplex=True, **kwargs):
        """MAP estimate by Nelder-Mead over :meth:`log_likelihood` (spectrum_model.py:635-696).  ``kwargs`` go to
        ``scipy.optimize.minimize`` as in the reference.

        The reference's loop is ~10^3 SERIAL scalar evaluations.  Here (``batch_simplex=True``, a plain Nelder-Mead run:
        no other ``method``, no bounds) the candidate points of every simplex iteration -- reflection, expansion, both
        contractions: all functions of the current simplex -- are evaluated as ONE device batch of four, the initial
        simplex and every shrink step as one batch each (:mod:`starfish_amd._neldermead`); the accept / contract / shrink
        decisions, their order, ``nit`` / ``nfev`` / ``status`` and the exceptions of invalid points are scipy's.  The
        state the model is left in is the reference's too: the parameters, caches, ``residuals`` entry and
        ``_lnprob`` of the LAST point the objective was asked for, then ``soln.x`` if the run succeeded."""
        from scipy.optimize import minimize

        from .._neldermead import minimize_neldermead_batched, split_minimize_kwargs

        priors = {} if priors is None else priors
        for key, val in priors.items():
            if key not in self.params and not key.startswith("cheb"):
                raise ValueError(f"Invalid priors: {key!r} is not a parameter of this model")
            if not callable(getattr(val, "logpdf", None)):
                raise ValueError(f"Invalid priors. {key} does not have a `logpdf` method")
            log_prob = val.logpdf(self[key])
            if not np.isfinite(log_prob):
                raise RuntimeError(f"{key}'s logpdf evaluated to {log_prob}")

        def nll(P):
            self.set_param_vector(P)
            return -self.log_likelihood(priors)

        nm_opts, why_not = split_minimize_kwargs(kwargs) if batch_simplex else (None, "batch_simplex=False")
        if nm_opts is None:
            self.log.debug(f"train: serial scipy path ({why_not})")
            opts = {"method": "Nelder-Mead"}
            opts.update(kwargs)
            soln = minimize(nll, self.get_param_vector(), **opts)
        else:
            def nll_batch(X):
                lnl, info = self.log_likelihood_batch(X, priors, return_info=True)

                def raiser(i):  # the scalar objective raises for these (spectrum_model.py:400, emulator.py:377-378, ...)
                    self._raise_for_info(info[i])

                return -lnl, raiser

            soln = minimize_neldermead_batched(nll_batch, self.get_param_vector(), **nm_opts)
            # leave the model where the reference's last objective call left it (parameters, frozen-cache snapshots, the
            # residual deque, _lnprob): ONE scalar evaluation at that point
            nll(soln.last_x)
        if soln.success:
            self.set_param_vector(soln.x)
        return soln

    # ------------------------------------------------------------------ persistence
    def save(self, filename, metadata=None):
        """Write parameters / frozen list / metadata as TOML (spectrum_model.py:592-619)."""
        from .._toml import dumps

        meta = {"name": self.name, "data": self.data_name}
        if self.emulator.name is not None:
            meta["emulator"] = self.emulator.name
        if metadata is not None:
            meta.update(metadata)
        doc = {"parameters": self.params.as_dict(), "frozen": list(self.frozen), "metadata": meta}
        with open(filename, "w") as fh:
            fh.write(dumps(doc))
        self.log.info(f"Saved current state at {filename}")

    def load(self, filename):
        """Read a state written by :meth:`save` (spectrum_model.py:621-633)."""
        from .._toml import load

        data = load(filename)
        self.params = FlatterDict(data["parameters"])
        self.frozen = list(data["frozen"])
        self._glob_snapshot = None
        self._loc_snapshot = None

    def plot(self, *args, **kwargs):
        raise NotImplementedError("plotting is out of scope for the MI355X hot path")

    def __repr__(self):
        out = f"{self.name}\n" + "-" * len(self.name) + "\n"
        out += f"Data: {self.data_name}\n"
        out += f"Emulator: {self.emulator.name}\n"
        out += f"Log Likelihood: {self._lnprob}\n"
        out += "\nParameters\n"
        for key, value in self.get_param_dict().items():
            if key == "global_cov":
                out += "  global_cov:\n"
                for gkey, gval in value.items():
                    out += f"    {gkey}: {gval}\n"
            elif key == "local_cov":
                out += "  local_cov:\n"
                kernels = value.values() if isinstance(value, dict) else value
                for i, kern in enumerate(kernels):
                    out += f"    {i}: " + ", ".join(f"{k}: {v}" for k, v in kern.items()) + "\n"
            elif key == "cheb":
                out += f"  cheb: {list(value.values())}\n"
            else:
                out += f"  {key}: {value}\n"
        if "log_scale" not in self.params and self._log_scale is not None:
            out += f"  log_scale: {self._log_scale} (fit)\n"
        if self.frozen:
            out += "\nFrozen Parameters\n"
            for key in self.frozen:
                if key in ("global_cov", "local_cov", "cheb"):
                    continue
                out += f"  {key}: {self[key]}\n"
        return out[:-1]
