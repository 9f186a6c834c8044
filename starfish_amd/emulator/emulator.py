"""
Query side of the Bayesian spectral emulator (reference: Starfish/emulator/emulator.py).

Kept: the constructor and its attributes, ``__call__`` (GP conditional mean / covariance of the PCA
weights), ``bulk_fluxes``, ``norm_factor``, hyper-parameter accessors, ``load`` / ``save``.
``__call__`` runs in the ``k_emu_*`` HIP kernels through ``sf_emulator_query_batch``; the constant
``v11`` is factored once per hyper-parameter set instead of on every call (emulator.py:387-388).
``log_likelihood`` / ``train`` reuse the batched Cholesky kernels.
Out of scope (one-time offline set-up): ``from_grid`` (PCA), plotting.
"""
import logging
import os
import warnings

import numpy as np

from .. import _device as D
from ..utils import calculate_dv
from ._utils import get_phi_squared
from .kernels import batch_kernel

log = logging.getLogger(__name__)


class Emulator:
    def __init__(
        self,
        grid_points,
        param_names,
        wavelength,
        weights,
        eigenspectra,
        w_hat,
        flux_mean,
        flux_std,
        factors,
        lambda_xi=1.0,
        variances=None,
        lengthscales=None,
        name=None,
    ):
        self.log = logging.getLogger(self.__class__.__name__)
        self.grid_points = np.asarray(grid_points, dtype=np.float64)
        self.param_names = param_names
        self.wl = np.asarray(wavelength, dtype=np.float64)
        self.weights = weights
        self.eigenspectra = np.asarray(eigenspectra, dtype=np.float64)
        self.flux_mean = np.asarray(flux_mean, dtype=np.float64)
        self.flux_std = np.asarray(flux_std, dtype=np.float64)
        self.factors = np.asarray(factors, dtype=np.float64)
        self._factor_interpolator = None

        self.dv = calculate_dv(self.wl)
        self.ncomps = self.eigenspectra.shape[0]

        self.hyperparams = {}
        self.name = name
        self.lambda_xi = lambda_xi
        self.variances = variances if variances is not None else 1e4 * np.ones(self.ncomps)

        unique = [sorted(np.unique(col)) for col in self.grid_points.T]
        self._grid_sep = np.array([np.diff(u).max() for u in unique])
        if lengthscales is None:
            lengthscales = np.tile(3 * self._grid_sep, (self.ncomps, 1))
        self.lengthscales = lengthscales

        self.min_params = self.grid_points.min(axis=0)
        self.max_params = self.grid_points.max(axis=0)

        self.iPhiPhi = np.linalg.inv(get_phi_squared(self.eigenspectra, self.grid_points.shape[0]))
        self.w_hat = np.asarray(w_hat, dtype=np.float64)
        self._trained = False
        self._device = None
        self._v11_assigned = False
        self._train_dev = None
        self._refresh_v11()

    # ----------------------------------------------------------------- hyper-parameters
    def _refresh_v11(self):
        """The hyper-parameters changed: v11 (emulator.py:126-128 / 569-571) is rebuilt on the HOST only when somebody
        reads it (queries, model contexts) -- the training objective builds and factors it on the device
        (:meth:`log_likelihood`), so a Nelder-Mead step does not pay the 17 ms numpy build of the 1320 x 1320 matrix
        of the worked example."""
        self._v11 = None
        self._v11_key = None
        self._v11_assigned = False
        self._device = None  # the device-side factor of v11 is rebuilt lazily
        self._factor = None

    def _hyper_key(self):
        return tuple(float(v) for v in self.hyperparams.values())

    @property
    def v11(self):
        # keyed on a snapshot of the hyper-parameter vector: the property setters (lambda_xi, variances, lengthscales)
        # and direct edits of ``hyperparams`` are seen by every consumer -- host matrix, query context, model contexts
        # and the device-built matrix of log_likelihood() always describe the same hyper-parameters.  A matrix assigned
        # by hand stays until set_param_dict / set_param_vector.
        if not self._v11_assigned and (self._v11 is None or self._v11_key != self._hyper_key()):
            self._device = None
            self._factor = None
            self._v11_key = self._hyper_key()
            self._v11 = self.iPhiPhi / self.lambda_xi + batch_kernel(
                self.grid_points, self.grid_points, self.variances, self.lengthscales
            )
        return self._v11

    @v11.setter
    def v11(self, value):
        self._v11 = None if value is None else np.asarray(value, dtype=np.float64)
        self._device = None
        self._factor = None
        self._v11_assigned = value is not None  # an explicitly assigned matrix is what log_likelihood factors

    def v11_factor(self):
        """(Linv, alpha) of the current v11, computed once per hyper-parameter set and handed to every order
        context built on this emulator."""
        if self._factor is None or self._factor[0] is not self.v11 or self._factor[1] is not self.w_hat:
            # keyed on the arrays themselves: any reassignment of v11 / w_hat invalidates the factor
            self._factor = (self.v11, self.w_hat, D.factor_v11(self.v11, self.w_hat))
        return self._factor[2]

    @property
    def lambda_xi(self):
        return np.exp(self.hyperparams["log_lambda_xi"])

    @lambda_xi.setter
    def lambda_xi(self, value):
        self.hyperparams["log_lambda_xi"] = np.log(value)

    @property
    def variances(self):
        vals = [v for k, v in self.hyperparams.items() if k.startswith("log_variance:")]
        return np.exp(vals)

    @variances.setter
    def variances(self, values):
        for i, value in enumerate(values):
            self.hyperparams[f"log_variance:{i}"] = np.log(value)

    @property
    def lengthscales(self):
        vals = [v for k, v in self.hyperparams.items() if k.startswith("log_lengthscale:")]
        return np.exp(vals).reshape(self.ncomps, -1)

    @lengthscales.setter
    def lengthscales(self, values):
        for i, value in enumerate(values):
            for j, ls in enumerate(value):
                self.hyperparams[f"log_lengthscale:{i}:{j}"] = np.log(ls)

    def __getitem__(self, key):
        return self.hyperparams[key]

    def get_param_dict(self):
        return self.hyperparams

    def set_param_dict(self, params):
        for key, val in params.items():
            if key in self.hyperparams:
                self.hyperparams[key] = val
        self._refresh_v11()

    def get_param_vector(self):
        return np.array(list(self.get_param_dict().values()))

    def set_param_vector(self, params):
        parameters = self.get_param_dict()
        if len(params) != len(parameters):
            raise ValueError("params must match length of parameters (get_param_vector())")
        self.set_param_dict(dict(zip(parameters.keys(), params)))

    # ----------------------------------------------------------------- query
    def _dev(self):
        self.v11  # (drops a context built for other hyper-parameters)
        if self._device is None:
            z = np.zeros(0)
            self._device = D.DeviceOrder(
                z, z, z, z, np.zeros((0, 0)), self.grid_points, self.variances, self.lengthscales,
                self.v11, self.w_hat, emu_factor=self.v11_factor(),
            )
        return self._device

    def __call__(self, params, full_cov=True, reinterpret_batch=False):
        """mu, cov of the PCA weights at ``params`` (Starfish/emulator/emulator.py:330-394).

        A single parameter vector gives ``mu (m,)`` and ``cov (m, m)``.  With
        ``reinterpret_batch=True`` a list of vectors returns per-point means ``(B, m)`` and variances
        ``(B, m)``; without it several vectors give the joint conditional, component-major
        (``mu (B m,)``, ``cov (B m, B m)``) as the reference's block-diagonal ``batch_kernel`` orders it."""
        params = np.atleast_2d(np.asarray(params, dtype=np.float64))
        if full_cov and reinterpret_batch:
            raise ValueError("Cannot reshape the full_covariance matrix for many parameters.")
        if not self._trained:
            warnings.warn(
                "This emulator has not been trained and therefore is not reliable. call "
                "emulator.train() to train."
            )
        if np.any(params < self.min_params) or np.any(params > self.max_params):
            raise ValueError("Querying emulator outside of original parameter range.")
        if params.shape[0] > 1 and not reinterpret_batch:
            # joint conditional over all query points, component-major like the reference's batch_kernel
            mu, cov, info = self._dev().emulator_query_joint(params)
            if np.any(info != 0):
                raise ValueError(D.INFO_MESSAGES.get(int(info[info != 0][0]), "emulator query failed"))
            return (mu, cov) if full_cov else (mu, np.diag(cov))
        mu, cov, info = self._dev().emulator_query(params)
        if np.any(info != 0):
            raise ValueError(D.INFO_MESSAGES.get(int(info[info != 0][0]), "emulator query failed"))
        if reinterpret_batch:
            return mu.squeeze(), np.diagonal(cov, axis1=1, axis2=2).squeeze()
        mu, cov = mu[0], cov[0]
        if not full_cov:
            cov = np.diag(cov)
        return mu, cov

    @property
    def bulk_fluxes(self):
        """vstack of eigenspectra, flux_mean, flux_std (Starfish/emulator/emulator.py:396-402)."""
        return np.vstack([self.eigenspectra, self.flux_mean, self.flux_std])

    def norm_factor(self, params):
        """Linear interpolation of the library normalisation factors
        (Starfish/emulator/emulator.py:429-444; host-side, only used when ``norm=True``)."""
        if self._factor_interpolator is None:
            from scipy.interpolate import LinearNDInterpolator

            self._factor_interpolator = LinearNDInterpolator(self.grid_points, self.factors, rescale=True)
        return self._factor_interpolator(np.asarray(params))

    def load_flux(self, params, norm=False):
        """One random realisation of the emulated spectrum at ``params``: weights drawn from the GP
        conditional, reconstructed through the eigenspectra (Starfish/emulator/emulator.py:404-427).
        The random draw is numpy's (host), as in the reference; mean and covariance come from the device."""
        P = np.atleast_2d(np.asarray(params, dtype=np.float64))
        fluxes = np.empty((P.shape[0], self.eigenspectra.shape[-1]))
        X = self.eigenspectra * self.flux_std
        for i, p in enumerate(P):
            mu, cov = self(p)
            fluxes[i] = np.random.multivariate_normal(mu, cov) @ X + self.flux_mean
        if norm:
            fluxes *= np.atleast_1d(self.norm_factor(P))[:, np.newaxis]
        return np.squeeze(fluxes)

    def determine_chunk_log(self, wavelength, buffer=50):
        """Truncate ``wl`` / ``eigenspectra`` to the shortest power-of-two-length window of the emulator grid
        that still covers ``wavelength`` +- ``buffer`` Angstrom (Starfish/emulator/emulator.py:446-482;
        the window selection follows Starfish/grid_tools/utils.py:181-240)."""
        wavelength = np.asarray(wavelength, dtype=np.float64)
        wl_min, wl_max = wavelength.min() - buffer, wavelength.max() + buffer
        wl = self.wl
        if wl_min < wl.min() or wl_max > wl.max():
            raise AssertionError(
                f"determine_chunk_log: wl_min {wl_min:.2f} and wl_max {wl_max:.2f} are not within the bounds "
                f"of the grid {wl.min():.2f} to {wl.max():.2f}."
            )
        inside = np.nonzero((wl >= wl_min) & (wl <= wl_max))[0]
        chunk = len(wl)
        while chunk // 2 > inside.size:
            chunk //= 2
        if chunk < len(wl):
            # the reference centres the window on the grid point nearest to the middle of the requested range
            # (grid_tools/utils.py:223-227) and asserts that it covers the range; where that window would leave
            # the grid or cut an end off, it is slid instead of failing
            centre = int(np.abs(wl - (wl_min + wl_max) / 2.0).argmin())
            lo = min(max(centre - chunk // 2, 0), len(wl) - chunk)
            lo = min(lo, inside[0])
            lo = max(lo, inside[-1] + 1 - chunk)
            sel = slice(lo, lo + chunk)
            self.wl = wl[sel]
            self.eigenspectra = self.eigenspectra[:, sel]
            self.flux_mean = self.flux_mean[sel]
            self.flux_std = self.flux_std[sel]
            self._device = None  # static device data is rebuilt on the next query
        assert self.wl.min() <= wl_min and self.wl.max() >= wl_max

    def get_index(self, params):
        params = np.atleast_2d(params)
        marks = np.abs(self.grid_points - np.expand_dims(params, 1)).sum(axis=-1)
        return marks.argmin(axis=1).squeeze()

    # ----------------------------------------------------------------- persistence
    @classmethod
    def load(cls, filename):
        """HDF5 layout of Starfish/emulator/emulator.py:188-231 (needs h5py); a file name ending in ``.npz`` is
        read as the numpy container `save` writes when h5py is not installed (same keys)."""
        filename = os.path.expandvars(filename)
        if str(filename).endswith(".npz"):
            with np.load(filename, allow_pickle=False) as base:
                kw = {k: base[k] for k in ("grid_points", "wavelength", "weights", "eigenspectra", "flux_mean", "flux_std",
                                           "w_hat", "factors", "variances", "lengthscales")}
                kw["param_names"] = [str(n) for n in base["param_names"]]
                kw["lambda_xi"] = float(base["lambda_xi"])
                trained = bool(base["trained"])
                name = str(base["name"]) if "name" in base.files else ".".join(str(filename).split(".")[:-1])
            emu = cls(name=name, **kw)
            emu._trained = trained
            return emu
        try:
            import h5py
        except ImportError as e:  # pragma: no cover - h5py is absent on the GPU box
            raise ImportError("Emulator.load needs h5py; construct the Emulator from arrays instead") from e
        filename = os.path.expandvars(filename)
        with h5py.File(filename, "r") as base:
            kw = dict(
                grid_points=base["grid_points"][:],
                param_names=base["grid_points"].attrs["names"],
                wavelength=base["wavelength"][:],
                weights=base["weights"][:],
                eigenspectra=base["eigenspectra"][:],
                flux_mean=base["flux_mean"][:],
                flux_std=base["flux_std"][:],
                w_hat=base["w_hat"][:],
                factors=base["factors"][:],
                lambda_xi=base["hyperparameters"]["lambda_xi"][()],
                variances=base["hyperparameters"]["variances"][:],
                lengthscales=base["hyperparameters"]["lengthscales"][:],
            )
            trained = base.attrs["trained"]
            name = base.attrs["name"] if "name" in base.attrs else ".".join(filename.split(".")[:-1])
        emu = cls(name=name, **kw)
        emu._trained = trained
        return emu

    def save(self, filename):
        """HDF5 layout of Starfish/emulator/emulator.py:233-270 (needs h5py); ``*.npz``: the same keys in a numpy
        container (no h5py needed)."""
        filename = os.path.expandvars(filename)
        if str(filename).endswith(".npz"):
            extra = {} if self.name is None else {"name": np.array(self.name)}
            np.savez_compressed(
                filename, grid_points=self.grid_points, param_names=np.array(list(self.param_names)), wavelength=self.wl,
                weights=self.weights, eigenspectra=self.eigenspectra, flux_mean=self.flux_mean, flux_std=self.flux_std,
                w_hat=self.w_hat, factors=self.factors, lambda_xi=np.array(self.lambda_xi), variances=self.variances,
                lengthscales=self.lengthscales, trained=np.array(bool(self._trained)), **extra)
            return
        try:
            import h5py
        except ImportError as e:  # pragma: no cover
            raise ImportError("Emulator.save needs h5py") from e
        filename = os.path.expandvars(filename)
        with h5py.File(filename, "w") as base:
            gp = base.create_dataset("grid_points", data=self.grid_points, compression=9)
            gp.attrs["names"] = self.param_names
            base.create_dataset("wavelength", data=self.wl, compression=9)
            base.create_dataset("weights", data=self.weights, compression=9)
            base.create_dataset("eigenspectra", data=self.eigenspectra, compression=9)
            base.create_dataset("flux_mean", data=self.flux_mean, compression=9)
            base.create_dataset("flux_std", data=self.flux_std, compression=9)
            base.create_dataset("w_hat", data=self.w_hat, compression=9)
            base.attrs["trained"] = self._trained
            if self.name is not None:
                base.attrs["name"] = self.name
            base.create_dataset("factors", data=self.factors, compression=9)
            hp = base.create_group("hyperparameters")
            hp.create_dataset("lambda_xi", data=self.lambda_xi)
            hp.create_dataset("variances", data=self.variances, compression=9)
            hp.create_dataset("lengthscales", data=self.lengthscales, compression=9)

    @classmethod
    def from_grid(cls, grid, **pca_kwargs):
        raise NotImplementedError("Emulator.from_grid (PCA of a spectral library) is offline set-up, out of scope")

    def log_likelihood(self, _retry=True):
        """-(logdet v11 + w_hat^T v11^-1 w_hat) / 2  (Starfish/emulator/emulator.py:602-619), entirely on the device:
        v11 is built from the hyper-parameters by ``sf_emulator_v11_build`` (grid, iPhiPhi and w_hat stay resident), then
        factored and solved by the same batched HIP kernels as the spectrum likelihood (``sf_potrf_batch`` /
        ``sf_logdet_sqmah_batch``).  One small upload (the hyper-parameters) and one small download
        per call.  A matrix assigned to ``self.v11`` by hand is uploaded and factored as it is."""
        from .. import _lib

        lib = _lib.require_gpu()
        torch = D._torch()
        dev = D.device_of()
        M, P = self.grid_points.shape
        m = self.ncomps
        n = m * M
        npad = -(-n // 64) * 64
        lda = npad + 16
        td = self._train_dev
        # resident training state, keyed on the identity of the arrays it was uploaded from: reassigning grid_points,
        # iPhiPhi or w_hat re-uploads; after an IN-PLACE edit of one of them set ``emu._train_dev = None``
        if (td is None or td["dev"] != dev or td["w_hat"] is not self.w_hat or td["grid_src"] is not self.grid_points
                or td["iphiphi_src"] is not self.iPhiPhi):
            R = np.zeros(npad)
            R[:n] = self.w_hat
            td = self._train_dev = dict(
                dev=dev, w_hat=self.w_hat, grid_src=self.grid_points, iphiphi_src=self.iPhiPhi,
                grid=D.to_dev(self.grid_points, dev), iphiphi=D.to_dev(self.iPhiPhi, dev),
                R=D.to_dev(R, dev), A=D.empty((npad, lda), dev), out=D.empty((2,), dev), info=D.empty((1,), dev, torch.int32),
                ws=D.workspace(lib.sf_potrf_workspace_bytes(npad, 1), dev),
            )
        s = D.stream_ptr(dev)
        A = td["A"]
        if self._v11_assigned:
            host = np.zeros((npad, lda))
            host[:n, :n] = self._v11
            idx = np.arange(n, npad)
            host[idx, idx] = 1.0  # identity padding: log 1 = 0, zero right-hand side
            A.copy_(torch.from_numpy(host))
        else:
            hyper = D.to_dev(np.concatenate([[self.lambda_xi], self.variances, np.asarray(self.lengthscales).ravel()]), dev)
            _lib.check(lib.sf_emulator_v11_build(D.ptr(td["grid"]), M, P, m, D.ptr(hyper), D.ptr(td["iphiphi"]), D.ptr(A),
                                                 npad, lda, s), "sf_emulator_v11_build")
        ws = td["ws"]
        _lib.check(lib.sf_potrf_batch(D.ptr(A), npad, lda, npad * lda, 1, D.ptr(td["info"]), D.ptr(ws), ws.numel(), s),
                   "sf_potrf_batch")
        _lib.check(lib.sf_logdet_sqmah_batch(D.ptr(A), npad, lda, npad * lda, 1, D.ptr(td["R"]), npad, D.ptr(ws),
                                             ws.numel(), D.ptr(td["out"][0:1]), D.ptr(td["out"][1:2]), s), "sf_logdet_sqmah_batch")
        code = int(td["info"].cpu()[0])
        if code == D.INFO_INTERNAL:
            if not _retry:
                raise RuntimeError(D.INFO_MESSAGES[D.INFO_INTERNAL])
            D.recover_from_internal(lib, "Emulator.log_likelihood", 1)
            return self.log_likelihood(_retry=False)  # (A was overwritten: it is rebuilt / re-uploaded above)
        if code != 0:
            raise np.linalg.LinAlgError(f"{code}-th leading minor of the array is not positive definite")
        ld, sq = td["out"].cpu().tolist()
        return -(ld + sq) / 2

    def train(self, **opt_kwargs):
        """Nelder-Mead over the hyper-parameter vector (Starfish/emulator/emulator.py:484-524); every
        likelihood evaluation factors v11 on the GPU."""
        from scipy.optimize import minimize

        def nll(P):
            if np.any(~np.isfinite(P)):
                return np.inf
            self.set_param_vector(P)
            if np.any(self.lengthscales < 2 * self._grid_sep):
                return np.inf
            return -self.log_likelihood()

        kwargs = {"method": "Nelder-Mead", "options": {"maxiter": 10000}}
        kwargs.update(opt_kwargs)
        soln = minimize(nll, self.get_param_vector(), **kwargs)
        if not soln.success:
            self.log.warning("Optimization did not succeed.")
            self.log.info(soln.message)
        else:
            self.set_param_vector(soln.x)
            self._trained = True
            self.log.info("Finished optimizing emulator hyperparameters")
        return soln

    def __repr__(self):
        out = "Emulator\n" + "-" * 8 + "\n"
        if self.name is not None:
            out += f"Name: {self.name}\n"
        out += f"Trained: {self._trained}\n"
        out += f"lambda_xi: {self.lambda_xi:.3f}\n"
        out += "Variances:\n" + "\n".join(f"\t{v:.2f}" for v in self.variances)
        out += "\nLengthscales:\n" + "\n".join(
            "\t[ " + " ".join(f"{l:.2f} " for l in ls) + "]" for ls in self.lengthscales
        )
        return out + "\n"
