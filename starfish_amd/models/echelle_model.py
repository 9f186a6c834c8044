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
    def __init__(self, emulator, data, grid_params, devices=None, name="EchelleModel", solver="dense", **params):
        """``params`` are shared by every order (vz, vsini, log_scale, global_cov, cheb, ...);
        per-order overrides can be set afterwards on ``self.orders[i]``.  ``solver`` as in
        :class:`SpectrumModel`: "dense" (the reference's algorithm: all (order x walker) units in one batched
        Cholesky), "auto" / "banded" (band + rank-m Woodbury per order, same value to rounding)."""
        self.name = name
        self.orders = []
        for i, order in enumerate(data):
            single = Spectrum(order._wave, order._flux, order._sigma, order.mask, name=f"{data.name}[{i}]")
            dev = None if devices is None else devices[i % len(devices)]
            kw = {k: (dict(v) if isinstance(v, dict) else ([dict(x) for x in v] if k == "local_cov" else
                       (list(v) if isinstance(v, (list, tuple)) else v))) for k, v in params.items()}
            self.orders.append(SpectrumModel(emulator, single, grid_params, device=dev, name=f"{name}[{i}]", solver=solver, **kw))

    @classmethod
    def from_orders(cls, models, name="EchelleModel"):
        """Assemble from existing per-order :class:`SpectrumModel` objects (e.g. one emulator chunk per order,
        per-order frozen local kernels).  All orders must expose the same thawed labels."""
        self = cls.__new__(cls)
        self.name = name
        self.orders = list(models)
        if not self.orders:
            raise ValueError("EchelleModel needs at least one order")
        labels = self.orders[0].labels
        for m in self.orders[1:]:
            if m.labels != labels:
                raise ValueError(f"orders disagree on the thawed parameters: {m.labels} vs {labels}")
        return self

    def __len__(self):
        return len(self.orders)

    def _side_streams(self, device, n=4):
        import torch

        pools = self.__dict__.setdefault("_streams", {})
        key = str(device)
        if key not in pools:
            pools[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
        return pools[key]

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

    def log_likelihood_batch(self, P, priors=None, return_info=False, return_orders=False):
        """lnL (B,) for B shared parameter vectors.  All (order x walker) units of a device are evaluated in ONE
        enqueue (``sf_loglike_multi_batch``: every order fills its own covariance matrices, all of them share one
        batched Cholesky) with one host synchronisation per device; devices work concurrently.  The sum over
        orders happens on the host (no collective).  Walkers that fail in any order get ``-inf``; ``info`` is the
        first non-zero per-order code.  With ``return_orders`` the (n_orders, B) per-order values are returned too."""
        from .. import _device as D

        P = np.atleast_2d(np.asarray(P, dtype=np.float64))
        B = P.shape[0]
        first = self.orders[0]
        prior_lp = first._batch_prior(P, priors)
        finite = np.isfinite(prior_lp)
        lnl = np.full(B, -np.inf)
        info = np.zeros(B, dtype=np.int32)
        per_order = np.full((len(self.orders), B), -np.inf)
        structured = any(m.solver != "dense" for m in self.orders)
        if finite.any() and structured:
            # structure-exploiting solver: one banded call per order (a few ms each instead of a share of the dense
            # batch; orders may need different half-widths, so they are not merged)
            vals = np.zeros((len(self.orders), int(finite.sum())))
            codes = np.zeros((len(self.orders), int(finite.sum())), dtype=np.int32)
            # every order is enqueued before anything is waited for; the orders of a device take turns on a few
            # side streams (a banded call of 64 walkers fills half of the chip) -- ONE synchronisation per device
            import torch

            pending, used = [], {}
            for i, m in enumerate(self.orders):
                dev, md, rows = m._pack(P[finite], update_caches=False)
                if m.solver == "dense":
                    pending.append((i, dev, md, None, rows))
                    continue
                pool = self._side_streams(dev.dev)
                st = pool[len(used.setdefault(str(dev.dev), [])) % len(pool)]
                used[str(dev.dev)].append(st)
                st.wait_stream(torch.cuda.current_stream(dev.dev))
                with torch.cuda.stream(st):
                    pending.append((i, dev, md, dev.structured_enqueue(md, rows), rows))
            for key in used:
                torch.cuda.synchronize(torch.device(key))
            for i, dev, md, pend, rows in pending:
                out = dev.loglike(md, rows) if pend is None else dev.structured_collect(md, pend, self.orders[i].solver)
                vals[i] = np.where(out["info"] == 0, out["lnl"], -np.inf)
                codes[i] = out["info"]
            per_order[:, finite] = vals
            lnl[finite] = vals.sum(axis=0) + prior_lp[finite]
            bad = codes != 0
            info[finite] = np.where(bad.any(axis=0), codes[bad.argmax(axis=0), np.arange(codes.shape[1])], 0)
        elif finite.any():
            packed = [m._pack(P[finite], update_caches=False) for m in self.orders]
            # one multi-order call per (device, row layout): the C-ABI reads every segment of a call with ONE
            # ModelDesc (row stride, offsets of the local kernels / Chebyshev terms, has_* flags), so orders whose
            # descriptors differ -- e.g. a different number of frozen local kernels -- go to separate calls
            groups = {}
            for idx, (dev, md, rows) in enumerate(packed):
                key = (str(dev.dev), dev.m, dev.P, int(np.atleast_2d(rows).shape[1])) + D.model_desc_key(md)
                groups.setdefault(key, []).append(idx)
            pending = []
            for idxs in groups.values():  # enqueue everything first, synchronise afterwards
                devs = [packed[i][0] for i in idxs]
                md = packed[idxs[0]][1]
                pending.append((idxs, D.loglike_multi(devs, md, [packed[i][2] for i in idxs], sync=False)))
            vals = np.zeros((len(self.orders), int(finite.sum())))
            codes = np.zeros((len(self.orders), int(finite.sum())), dtype=np.int32)
            for idxs, plan in pending:
                for i, out in zip(idxs, plan.collect()):
                    vals[i] = out["lnl"]
                    codes[i] = out["info"]
            per_order[:, finite] = vals
            lnl[finite] = vals.sum(axis=0) + prior_lp[finite]
            bad = codes != 0
            info[finite] = np.where(bad.any(axis=0), codes[bad.argmax(axis=0), np.arange(codes.shape[1])], 0)
        self.last_info = info
        out = (lnl,)
        if return_info:
            out += (info,)
        if return_orders:
            out += (per_order,)
        return out if len(out) > 1 else lnl
