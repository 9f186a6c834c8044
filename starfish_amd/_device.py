"""
Device-side plumbing: PyTorch-ROCm tensors own the HBM buffers and the stream, the arithmetic is done
by the HIP kernels behind the C-ABI (``_lib``).  Nothing here computes on the CPU.
"""

import ctypes as C

import numpy as np

from . import _lib

INFO_MESSAGES = {
    -1: "Querying emulator outside of original parameter range.",
    -2: "vsini must be positive",
    -3: "emulator weight covariance is not positive definite",
    -4: "covariance support wider than the band half-width given to the banded solver",
    -5: "internal error: a bounded wait inside a persistent kernel gave up (banded sweep or dataflow Cholesky; "
        "no result of that call is valid -- is the GPU shared with another process?)",
    -6: "the log-likelihood evaluated to NaN (non-finite input or intermediate)",
}
INFO_BANDWIDTH = -4
INFO_INTERNAL = -5
C_KMS = 2.99792458e5


def persistent_status(lib):
    """``sf_persistent_potrf_status`` as a dict (host memory of the library: read it after the stream was synchronised)."""
    buf = (C.c_longlong * 8)()
    _lib.check(lib.sf_persistent_potrf_status(buf), "sf_persistent_potrf_status")
    keys = ("aborted_launches", "reason", "workgroups_started", "grid", "wait_ticks", "tasks_completed",
            "launches", "enabled")
    return dict(zip(keys, (int(v) for v in buf)))


def recover_from_internal(lib, where, count):
    """A dense call came back with SF_INFO_INTERNAL: the persistent-kernel Cholesky gave a launch up and the whole batch
    is invalid (include/starfish_amd.h).  The reference would never turn that into a rejected proposal
    (spectrum_model.py:400 raises out of cho_factor), so: say so loudly -- with what the aborting workgroup recorded --,
    switch the PROCESS (all devices, all threads) to the launch sequences (no waits inside kernels) and let the caller
    re-run the batch."""
    import warnings

    st = persistent_status(lib)
    why = {1: "a wait between workgroups reached its 4-s bound",
           2: "no task of the launch completed for 25 ms"}.get(st["reason"], "no abort record")
    if st["grid"] and st["workgroups_started"] < st["grid"]:
        why += f"; only {st['workgroups_started']} of its {st['grid']} workgroups had started (grid not co-resident)"
    warnings.warn(
        f"{where}: internal status -5 for {int(count)} unit(s) -- the persistent-kernel Cholesky gave a launch up ({why}, "
        f"after {st['tasks_completed']} completed tasks): is this GPU shared with another process?  The kernel assumes "
        "one process per GPU.  It is now disabled for this whole process, on every device and thread "
        "(sf_persistent_potrf(0)), and the batch is re-run on the launch sequence.",
        RuntimeWarning,
        stacklevel=3,
    )
    lib.sf_persistent_potrf(0)


def _torch():
    import torch

    return torch


def device_of(index=None):
    torch = _torch()
    if index is None:
        index = torch.cuda.current_device() if torch.cuda.is_available() else 0
    return torch.device("cuda", index)


def to_dev(arr, dev):
    torch = _torch()
    a = np.ascontiguousarray(np.asarray(arr, dtype=np.float64))
    return torch.from_numpy(a).to(dev)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream_ptr(dev):
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def empty(shape, dev, dtype=None):
    torch = _torch()
    return torch.empty(shape, dtype=dtype or torch.float64, device=dev)


def workspace(nbytes, dev):
    torch = _torch()
    return torch.empty(max(int(nbytes), 8), dtype=torch.uint8, device=dev)


def band_halfwidth_bound(wave, rows, n_grid, has_global, n_local, n_cheb):
    """Per-walker upper bound on max|i-j| over the non-zero entries of the structured part of the
    covariance (global Matern taper r0 = 6 ls, Starfish/models/kernels.py:29; local patches r0 = 4 sigma,
    kernels.py:73), from host copies of the wavelength grid and of the C-ABI parameter rows
    (include/starfish_amd.h: [4] log_amp, [5] log_ls, locals after the grid and Chebyshev entries).
    Conservative: uses the smallest pixel spacing.  Pure host logic (numpy)."""
    w = np.asarray(wave, dtype=np.float64)
    rows = np.atleast_2d(np.asarray(rows, dtype=np.float64))
    big = np.iinfo(np.int32).max
    if w.size < 2 or not np.all(np.diff(w) > 0):
        return np.full(rows.shape[0], big, dtype=np.int64)
    hw = np.zeros(rows.shape[0])
    if has_global:
        # metric of the global kernel (kernels.py:27): r = c/2 |wi - wj| / (wi + wj), a quarter of the
        # velocity separation; r(i, i+d) >= d * dv * (1 - O(r0/c)) (slightly sub-additive)
        dv = float(np.min(C_KMS / 2 * (w[1:] - w[:-1]) / (w[1:] + w[:-1])))
        r0 = 6 * np.exp(rows[:, 5])
        hw = np.maximum(hw, np.floor(r0 / dv * (1 + 4 * r0 / C_KMS + 1e-9)) + 1)
    off = 6 + n_grid + n_cheb
    for k in range(n_local):
        mu = rows[:, off + 3 * k]
        r0 = 4 * np.exp(rows[:, off + 3 * k + 2])
        # metric d_i = c/mu |w_i - mu| (kernels.py:69): patch = pixels with d_i <= r0; its extent in
        # pixels is at most 2 r0 / (smallest step of d), step of d >= (c/mu) * min(diff(w))
        step = C_KMS / np.abs(mu) * float(np.min(np.diff(w)))
        hw = np.maximum(hw, np.floor(2 * r0 / step * (1 + 1e-9)) + 1)
    return np.minimum(hw, big).astype(np.int64)


def factor_v11(v11, w_hat):
    """Init-time constants of the emulator conditional (the reference solves with the constant v11 on every call,
    emulator.py:387-388): Linv = inverse of the lower Cholesky factor of v11 and alpha = v11^-1 w_hat, by LAPACK on
    the host, once per hyper-parameter set -- every order context of the emulator receives the same arrays."""
    from scipy.linalg import cho_solve, cholesky, solve_triangular

    try:
        L = cholesky(np.asarray(v11, dtype=np.float64), lower=True)
    except np.linalg.LinAlgError as e:  # the library's documented status for this case (SF_INFO_EMULATOR_NOT_PD)
        raise np.linalg.LinAlgError(INFO_MESSAGES[-3]) from e
    linv = np.tril(solve_triangular(L, np.eye(L.shape[0]), lower=True))
    alpha = cho_solve((L, True), np.asarray(w_hat, dtype=np.float64))
    return np.ascontiguousarray(linv), np.ascontiguousarray(alpha)


def model_desc_key(md):
    """The fields of a ModelDesc as a hashable tuple (two orders may share a multi-order call only if equal)."""
    return tuple(int(getattr(md, name)) for name, _ in md._fields_)


class MultiPlan:
    """Device-resident state of a repeated multi-order evaluation (sf_loglike_multi_batch): parameter rows,
    outputs and workspace are allocated once; :meth:`enqueue` only launches.  ``orders`` are
    :class:`DeviceOrder` objects of ONE device, ``rows_list[i]`` the (B_i, stride) C-ABI rows of order i.
    Unit lists that do not fit the free HBM (or ``max_units``) are cut into several calls."""

    def __init__(self, orders, md, rows_list, max_units=None):
        torch = _torch()
        self.orders = list(orders)
        self.md = md
        self.lib = orders[0].lib
        self.dev = orders[0].dev
        if any(o.dev != self.dev for o in orders):
            raise ValueError("multi-order call: all orders must live on the same device")
        stride = orders[0].param_stride(md)
        for o, r in zip(orders, rows_list):
            if o.param_stride(md) != stride or int(r.shape[-1]) != stride:
                raise ValueError(f"multi-order call: parameter rows of {int(r.shape[-1])} doubles do not match the "
                                 f"descriptor's stride {stride} (orders with different descriptors need separate calls)")
        self.sizes = [int(np.atleast_2d(r).shape[0]) if not torch.is_tensor(r) else int(r.shape[0]) for r in rows_list]
        U = sum(self.sizes)
        with torch.cuda.device(self.dev):
            self.P = [r if torch.is_tensor(r) else to_dev(np.atleast_2d(r), self.dev) for r in rows_list]
            self.quad = empty((4, U), self.dev)  # lnl, logdet, sqmah, log_scale: one device->host copy
            self.info = empty((U,), self.dev, torch.int32)
            one = _lib.Segment(orders[0].ctx, ptr(self.P[0]).value, 1, 0)
            # units that fit: the workspace is linear in the unit count up to the fixed Cholesky scratch
            w1 = self.lib.sf_multi_workspace_bytes(C.byref(one), 1, C.byref(md))
            one.B = 2
            w2 = self.lib.sf_multi_workspace_bytes(C.byref(one), 1, C.byref(md))
            per_unit = max(w2 - w1, 1)
            free, _total = torch.cuda.mem_get_info(self.dev)
            lead = orders[0]
            if lead._ws_multi is not None:
                free += lead._ws_multi.numel()
            cap = max(1, (int(free * 0.85) - (w1 - per_unit)) // per_unit)
            if max_units:
                cap = min(cap, int(max_units))
            self.pieces, cur, cur_n = [], [], 0
            for i, n in enumerate(self.sizes):
                lo = 0
                while lo < n:
                    take = min(n - lo, cap - cur_n)
                    cur.append((i, lo, lo + take))
                    cur_n += take
                    lo += take
                    if cur_n == cap:
                        self.pieces.append(cur)
                        cur, cur_n = [], 0
            if cur:
                self.pieces.append(cur)
            offs = np.concatenate([[0], np.cumsum(self.sizes)])
            self.calls, need = [], 0
            for piece in self.pieces:
                segs = (_lib.Segment * len(piece))()
                for k, (i, lo, hi) in enumerate(piece):
                    segs[k] = _lib.Segment(orders[i].ctx, ptr(self.P[i][lo:hi]).value, hi - lo, 0)
                nb = self.lib.sf_multi_workspace_bytes(segs, len(piece), C.byref(md))
                if nb == 0:
                    _lib.check(-1, "sf_multi_workspace_bytes")
                need = max(need, nb)
                u0 = int(offs[piece[0][0]] + piece[0][1])  # the pieces of one call are contiguous in unit order
                self.calls.append((segs, len(piece), u0, sum(hi - lo for _, lo, hi in piece)))
            if lead._ws_multi is None or lead._ws_multi.numel() < need:
                lead._ws_multi = None
                lead._ws_multi = workspace(need, self.dev)
            self.ws = lead._ws_multi

    @property
    def units(self):
        return sum(self.sizes)

    def enqueue(self):
        """Launch only: results land in ``self.quad`` / ``self.info`` once the device's current stream is done."""
        torch = _torch()
        with torch.cuda.device(self.dev):
            s = stream_ptr(self.dev)
            q = self.quad
            for segs, nseg, u0, n in self.calls:
                rc = self.lib.sf_loglike_multi_batch(
                    segs, nseg, C.byref(self.md), ptr(q[0][u0:u0 + n]), ptr(q[1][u0:u0 + n]), ptr(q[2][u0:u0 + n]),
                    ptr(q[3][u0:u0 + n]), ptr(self.info[u0:u0 + n]), ptr(self.ws), self.ws.numel(), s,
                )
                _lib.check(rc, "sf_loglike_multi_batch")

    def collect(self):
        out = collect_multi(self.quad, self.info, self.sizes)
        bad = sum(int((o["info"] == INFO_INTERNAL).sum()) for o in out)
        if bad:  # an aborted persistent launch invalidates the whole call: re-run on the launch sequence, once
            recover_from_internal(self.lib, "sf_loglike_multi_batch", bad)
            self.enqueue()
            out = collect_multi(self.quad, self.info, self.sizes)
            if any((o["info"] == INFO_INTERNAL).any() for o in out):  # (never an ordinary per-unit status: nothing is valid)
                raise RuntimeError(INFO_MESSAGES[INFO_INTERNAL])
        return out


def loglike_multi(orders, md, rows_list, max_units=None, sync=True):
    """(order x walker) units of several orders of ONE device in one enqueue and one host synchronisation.
    Returns a list of dicts (lnl, logdet, sqmah, log_scale, info) per order; with ``sync=False`` the
    un-synchronised :class:`MultiPlan` (callers overlapping several devices call ``plan.collect()`` later)."""
    plan = MultiPlan(orders, md, rows_list, max_units=max_units)
    plan.enqueue()
    return plan.collect() if sync else plan


def collect_multi(quad, info, sizes):
    host = quad.cpu().numpy()
    hinfo = info.cpu().numpy()
    out, u = [], 0
    for n in sizes:
        out.append(dict(lnl=host[0, u:u + n], logdet=host[1, u:u + n], sqmah=host[2, u:u + n],
                        log_scale=host[3, u:u + n], info=hinfo[u:u + n]))
        u += n
    return out


class DeviceOrder:
    """One ``sf_ctx``: the static data of an order + emulator resident in HBM, and the batched calls."""

    def __init__(
        self,
        wave,
        flux,
        sigma,
        min_dv_wave,
        bulk_fluxes,
        grid_points,
        variances,
        lengthscales,
        v11,
        w_hat,
        device=None,
        emu_factor=None,
    ):
        """``emu_factor``: optional ``(Linv, alpha)`` of the constant v11 (see :func:`factor_v11`), shared by all
        orders of one emulator; without it the library factors v11 itself (scalar host code)."""
        self.lib = _lib.require_gpu()
        torch = _torch()
        self.dev = device_of(device)
        f8 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))  # noqa: E731
        self._keep = [
            f8(wave),
            f8(flux),
            f8(sigma),
            f8(min_dv_wave),
            f8(bulk_fluxes),
            f8(grid_points),
            f8(variances),
            f8(lengthscales),
            f8(v11),
            f8(w_hat),
        ]
        w, fl, sg, mdw, bulk, grid, var, ls, v11a, wh = self._keep
        self.n = int(w.shape[0])
        self.nf = int(mdw.shape[0])
        self.m = int(var.shape[0])
        self.M, self.P = (int(grid.shape[0]), int(grid.shape[1]))
        if self.n:
            assert bulk.shape == (self.m + 2, self.nf), bulk.shape
        assert v11a.shape == (self.m * self.M,) * 2
        d = _lib.OrderDesc()
        d.n, d.nf, d.m, d.n_grid, d.M = self.n, self.nf, self.m, self.P, self.M
        d.wave, d.flux, d.sigma = map(_lib.as_double_p, (w, fl, sg))
        d.min_dv_wave, d.bulk_fluxes = _lib.as_double_p(mdw), _lib.as_double_p(bulk)
        d.grid_points, d.variances = _lib.as_double_p(grid), _lib.as_double_p(var)
        d.lengthscales, d.v11, d.w_hat = map(_lib.as_double_p, (ls, v11a, wh))
        if emu_factor is not None:
            linv, alpha = f8(emu_factor[0]), f8(emu_factor[1])
            assert linv.shape == v11a.shape and alpha.shape == (v11a.shape[0],)
            self._keep += [linv, alpha]
            d.linv, d.alpha = _lib.as_double_p(linv), _lib.as_double_p(alpha)
        err = C.c_int(0)
        with torch.cuda.device(self.dev):
            self.ctx = self.lib.sf_ctx_create(C.byref(d), self.dev.index or 0, C.byref(err))
        if not self.ctx:
            _lib.check(err.value or -1, "sf_ctx_create")
        self.npad = self.lib.sf_ctx_npad(self.ctx)
        self.lda = self.lib.sf_ctx_lda(self.ctx)
        self._ws = None
        self._ws_multi = None  # workspace of loglike_multi calls led by this order

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.sf_ctx_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def model_desc(self, has_vsini, has_vz, has_log_scale, has_global, n_local, n_cheb, use_sigma_w=False,
                   has_av=False):
        md = _lib.ModelDesc()
        md.has_vsini, md.has_vz = int(has_vsini), int(has_vz)
        md.has_log_scale, md.has_global = int(has_log_scale), int(has_global)
        md.n_local, md.n_cheb, md.use_sigma_w = int(n_local), int(n_cheb), int(use_sigma_w)
        md.has_av = int(has_av)
        return md

    def param_stride(self, md):
        return self.lib.sf_param_stride(self.ctx, C.byref(md))

    def workspace_bytes(self, md, B):
        return self.lib.sf_workspace_bytes(self.ctx, C.byref(md), int(B))

    def max_batch(self, md, reserve_fraction=0.15):
        """Largest batch whose workspace fits the free HBM (leaving a safety margin)."""
        torch = _torch()
        free, _total = torch.cuda.mem_get_info(self.dev)
        if self._ws is not None:
            free += self._ws.numel()
        budget = int(free * (1.0 - reserve_fraction))
        per = self.workspace_bytes(md, 1)
        return max(1, budget // max(per, 1))

    def _reserve(self, need):
        """The order's workspace, grown to ``need`` bytes.  The buffer is handed to kernels on whatever stream is
        current (EchelleModel rotates orders over side streams), so that stream is recorded on it: the caching
        allocator then does not recycle a dropped buffer before the work queued on it has finished."""
        torch = _torch()
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = workspace(need, self.dev)
        self._ws.record_stream(torch.cuda.current_stream(self.dev))
        return self._ws

    def _work(self, md, B):
        return self._reserve(self.workspace_bytes(md, B))

    def release_workspace(self):
        self._ws = None
        self._ws_multi = None

    # ------------------------------------------------------------------ structure-exploiting solver
    def banded_window_halfwidth(self):
        """Half-widths up to this value use the LDS-window sweep; wider ones the in-place HBM/L2 kernel."""
        return int(self.lib.sf_banded_window_halfwidth(self.ctx)) if self.n else -1

    def banded_max_halfwidth(self):
        """Largest band half-width (pixels) sf_loglike_banded_batch accepts for this order; -1 = unusable."""
        return int(self.lib.sf_banded_max_halfwidth(self.ctx)) if self.n else -1

    def halfwidth_bound(self, md, rows):
        """Per-walker upper bound (pixels) on the support of the structured part of the covariance."""
        return band_halfwidth_bound(self._keep[0], rows, self.P, bool(md.has_global), int(md.n_local), int(md.n_cheb))

    def banded_workspace_bytes(self, md, B, halfwidth):
        return self.lib.sf_banded_workspace_bytes(self.ctx, C.byref(md), int(B), int(halfwidth))

    def _work_banded(self, md, B, halfwidth):
        return self._reserve(self.banded_workspace_bytes(md, B, halfwidth))

    def loglike_banded_device(self, md, P_dev, halfwidth, out_lnl, info=None, logdet=None, sqmah=None,
                              resid=None, log_scale=None):
        """Enqueue-only banded + rank-m solve (sf_loglike_banded_batch); device tensors in/out."""
        B = int(P_dev.shape[0])
        ws = self._work_banded(md, B, halfwidth)
        rc = self.lib.sf_loglike_banded_batch(
            self.ctx, C.byref(md), B, ptr(P_dev), int(halfwidth), ptr(out_lnl), ptr(logdet), ptr(sqmah),
            ptr(resid), ptr(log_scale), ptr(info), ptr(ws), ws.numel(), stream_ptr(self.dev),
        )
        _lib.check(rc, "sf_loglike_banded_batch")

    # ------------------------------------------------------------------ batched calls
    def loglike(self, md, params, want_resid=False, max_chunk=None, solver="dense", _retry=True):
        """params: (B, stride) float64 (numpy or cuda tensor) in the C-ABI row layout.
        Returns dict of numpy arrays: lnl, logdet, sqmah, log_scale, info (+ resid).

        solver: "dense"  -- the reference's algorithm, batched N x N Cholesky (sf_loglike_batch);
                "banded" -- band + rank-m Woodbury solve (sf_loglike_banded_batch); walkers whose
                            covariance support exceeds the window come back with info = -4;
                "auto"   -- banded for the walkers whose half-width bound fits, dense for the rest."""
        torch = _torch()
        if solver not in ("dense", "banded", "auto"):
            raise ValueError("solver must be 'dense', 'banded' or 'auto'")
        if solver != "dense":
            return self._loglike_structured(md, params, want_resid, max_chunk, solver)
        with torch.cuda.device(self.dev):
            P = params if torch.is_tensor(params) else to_dev(params, self.dev)
            B = int(P.shape[0])
            chunk = min(B, max_chunk or B, self.max_batch(md))
            quad = empty((4, B), self.dev)  # one device->host copy for the four double outputs
            lnl, logdet, sqmah, lsc = quad[0], quad[1], quad[2], quad[3]
            info = empty((B,), self.dev, torch.int32)
            resid = empty((B, self.n), self.dev) if want_resid else None
            s = stream_ptr(self.dev)
            for lo in range(0, B, chunk):
                hi = min(lo + chunk, B)
                ws = self._work(md, hi - lo)
                rc = self.lib.sf_loglike_batch(
                    self.ctx, C.byref(md), hi - lo, ptr(P[lo:hi]), ptr(lnl[lo:hi]), ptr(logdet[lo:hi]),
                    ptr(sqmah[lo:hi]), ptr(resid[lo:hi]) if want_resid else C.c_void_p(0), ptr(lsc[lo:hi]),
                    ptr(info[lo:hi]), ptr(ws), ws.numel(), s,
                )
                _lib.check(rc, "sf_loglike_batch")
            host = quad.cpu().numpy()
            out = dict(lnl=host[0], logdet=host[1], sqmah=host[2], log_scale=host[3], info=info.cpu().numpy())
            internal = out["info"] == INFO_INTERNAL
            if internal.any() and _retry:
                # never a silent -inf: warn, switch the persistent kernel off, evaluate the batch again
                recover_from_internal(self.lib, "sf_loglike_batch", internal.sum())
                return self.loglike(md, P, want_resid=want_resid, max_chunk=max_chunk, solver="dense", _retry=False)
            if want_resid:
                out["resid"] = resid.cpu().numpy()
            return out

    def _loglike_structured(self, md, params, want_resid, max_chunk, solver):
        return self.structured_collect(md, self.structured_enqueue(md, params, want_resid, max_chunk), solver)

    def structured_enqueue(self, md, params, want_resid=False, max_chunk=None):
        """First half of the structure-exploiting evaluation: group the walkers by covariance support and ENQUEUE the
        banded calls on the current stream -- no host synchronisation.  :meth:`structured_collect` (after the stream
        is done) fetches the results; callers with several orders enqueue them all first (EchelleModel)."""
        torch = _torch()
        rows = params.cpu().numpy() if torch.is_tensor(params) else np.asarray(params, dtype=np.float64)
        rows = np.atleast_2d(rows)
        B = rows.shape[0]
        wmax = self.banded_max_halfwidth()
        hw = self.halfwidth_bound(md, rows)
        fits = hw <= wmax  # (solver="banded": the walkers that do not fit are reported with info = -4)
        out = dict(
            lnl=np.full(B, -np.inf), logdet=np.full(B, np.nan), sqmah=np.full(B, np.nan),
            log_scale=np.full(B, np.nan), info=np.full(B, INFO_BANDWIDTH, dtype=np.int32),
        )
        if want_resid:
            out["resid"] = np.full((B, self.n), np.nan)
        groups = []
        # two groups: the cost of the wide-band factorisation grows with the half-width, so the walkers that fit the
        # LDS window are not dragged along with the wide ones
        wwin = self.banded_window_halfwidth()
        parts = [idx for idx in (np.nonzero(fits & (hw <= wwin))[0], np.nonzero(fits & (hw > wwin))[0]) if idx.size]
        # ONE allocation for both groups, made before anything is enqueued: the second group must not drop the
        # buffer the first group's kernels are still queued on
        if parts:
            self._reserve(max(self.banded_workspace_bytes(md, min(idx.size, max_chunk or idx.size), int(hw[idx].max()))
                              for idx in parts))
        for idx in parts:
            W = int(hw[idx].max())
            with torch.cuda.device(self.dev):
                P = to_dev(rows[idx], self.dev)
                nb = idx.size
                quad = empty((4, nb), self.dev)  # one device->host copy for the four double outputs
                lnl, logdet, sqmah, lsc = quad[0], quad[1], quad[2], quad[3]
                info = empty((nb,), self.dev, torch.int32)
                resid = empty((nb, self.n), self.dev) if want_resid else None
                chunk = min(nb, max_chunk or nb)
                for lo in range(0, nb, chunk):
                    hi = min(lo + chunk, nb)
                    self.loglike_banded_device(
                        md, P[lo:hi], W, lnl[lo:hi], info[lo:hi], logdet[lo:hi], sqmah[lo:hi],
                        resid[lo:hi] if want_resid else None, lsc[lo:hi],
                    )
                groups.append((idx, quad, info, resid, P))
        return dict(out=out, groups=groups, fits=fits, rows=rows, want_resid=want_resid, max_chunk=max_chunk)

    def structured_collect(self, md, pend, solver):
        out, rows, fits, want_resid = pend["out"], pend["rows"], pend["fits"], pend["want_resid"]
        for idx, quad, info, resid, _ in pend["groups"]:
            host = quad.cpu().numpy()
            for row, key in enumerate(("lnl", "logdet", "sqmah", "log_scale")):
                out[key][idx] = host[row]
            out["info"][idx] = info.cpu().numpy()
            if want_resid:
                out["resid"][idx] = resid.cpu().numpy()
        # an internal wait timeout of the sweep (-5, not expected: the bound keeps a logic error from hanging the GPU) is
        # recomputed by the dense solver -- under "banded" too -- instead of silently turning into a rejected walker
        internal = out["info"] == INFO_INTERNAL
        if internal.any():
            import warnings

            warnings.warn(f"banded solver: internal status -5 for {int(internal.sum())} walker(s); recomputed densely",
                          RuntimeWarning)
        rest = np.nonzero(internal)[0]
        if solver == "auto":  # too wide for the banded kernels -> dense
            rest = np.nonzero(~fits | (out["info"] == INFO_BANDWIDTH) | internal)[0]
        if rest.size:
            dense = self.loglike(md, rows[rest], want_resid=want_resid, max_chunk=pend["max_chunk"], solver="dense")
            for key in out:
                out[key][rest] = dense[key]
        return out

    def loglike_device(self, md, P_dev, out_lnl, info=None):
        """Enqueue-only variant for bench.py: device tensors in/out, no synchronisation."""
        B = int(P_dev.shape[0])
        ws = self._work(md, B)
        rc = self.lib.sf_loglike_batch(
            self.ctx, C.byref(md), B, ptr(P_dev), ptr(out_lnl), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0),
            C.c_void_p(0), ptr(info), ptr(ws), ws.numel(), stream_ptr(self.dev),
        )
        _lib.check(rc, "sf_loglike_batch")

    def forward(self, md, params):
        """SpectrumModel.__call__ for B rows: flux (B, n), cov (B, n, n), log_scale, info."""
        torch = _torch()
        with torch.cuda.device(self.dev):
            P = params if torch.is_tensor(params) else to_dev(params, self.dev)
            B = int(P.shape[0])
            flux = empty((B, self.n), self.dev)
            cov = empty((B, self.n, self.n), self.dev)
            lsc = empty((B,), self.dev)
            info = empty((B,), self.dev, torch.int32)
            ws = self._work(md, B)
            rc = self.lib.sf_forward_batch(
                self.ctx, C.byref(md), B, ptr(P), ptr(flux), ptr(cov), ptr(lsc), ptr(info), ptr(ws),
                ws.numel(), stream_ptr(self.dev),
            )
            _lib.check(rc, "sf_forward_batch")
            return dict(
                flux=flux.cpu().numpy(), cov=cov.cpu().numpy(), log_scale=lsc.cpu().numpy(),
                info=info.cpu().numpy(),
            )

    def cov_fill(self, md, params, ld=None, lower_only=False, add_jitter=False, guard=0):
        """The fused covariance fill alone (sf_cov_fill_batch): (B, n, ld) array, row stride ``ld`` >= n (columns
        beyond n are left as allocated: zero here).  Exactly n rows per matrix are written -- no identity padding, that
        belongs to the library's own workspace layout.  ``guard`` > 0 appends that many sentinel doubles (-7.0) behind the
        last matrix and returns them as a third value (tests: nothing may be written past the caller's array)."""
        torch = _torch()
        with torch.cuda.device(self.dev):
            P = params if torch.is_tensor(params) else to_dev(params, self.dev)
            B = int(P.shape[0])
            ld = int(ld or self.n)
            buf = torch.zeros((B * self.n * ld + int(guard),), dtype=torch.float64, device=self.dev)
            if guard:
                buf[B * self.n * ld:] = -7.0
            cov = buf[: B * self.n * ld].view(B, self.n, ld)
            info = empty((B,), self.dev, torch.int32)
            ws = self._work(md, B)
            rc = self.lib.sf_cov_fill_batch(
                self.ctx, C.byref(md), B, ptr(P), ptr(cov), ld, self.n * ld, int(lower_only), int(add_jitter), ptr(info),
                ptr(ws), ws.numel(), stream_ptr(self.dev),
            )
            _lib.check(rc, "sf_cov_fill_batch")
            if guard:
                return cov.cpu().numpy(), info.cpu().numpy(), buf[B * self.n * ld:].cpu().numpy()
            return cov.cpu().numpy(), info.cpu().numpy()

    def cov_fill_device(self, md, P_dev, cov, ld, stride, lower_only=False, add_jitter=True, info=None):
        """Enqueue-only sf_cov_fill_batch into a caller-held device array (bench.py's fill leg): no host copies."""
        torch = _torch()
        with torch.cuda.device(self.dev):
            B = int(P_dev.shape[0])
            ws = self._work(md, B)
            rc = self.lib.sf_cov_fill_batch(
                self.ctx, C.byref(md), B, ptr(P_dev), ptr(cov), int(ld), int(stride), int(lower_only), int(add_jitter),
                ptr(info) if info is not None else C.c_void_p(0), ptr(ws), ws.numel(), stream_ptr(self.dev),
            )
            _lib.check(rc, "sf_cov_fill_batch")

    def transform(self, md, params):
        torch = _torch()
        with torch.cuda.device(self.dev):
            P = params if torch.is_tensor(params) else to_dev(params, self.dev)
            B = int(P.shape[0])
            flux = empty((B, self.n), self.dev)
            X = empty((B, self.m, self.n), self.dev)
            resid = empty((B, self.n), self.dev)
            lsc = empty((B,), self.dev)
            info = empty((B,), self.dev, torch.int32)
            ws = self._work(md, B)
            rc = self.lib.sf_transform_batch(
                self.ctx, C.byref(md), B, ptr(P), ptr(flux), ptr(X), ptr(resid), ptr(lsc), ptr(info),
                ptr(ws), ws.numel(), stream_ptr(self.dev),
            )
            _lib.check(rc, "sf_transform_batch")
            return dict(
                flux=flux.cpu().numpy(), X=X.cpu().numpy(), resid=resid.cpu().numpy(),
                log_scale=lsc.cpu().numpy(), info=info.cpu().numpy(),
            )

    def emulator_query(self, grid_params):
        """grid_params: (B, P).  Returns mu (B, m), cov (B, m, m), info (B,)."""
        torch = _torch()
        gp = np.atleast_2d(np.asarray(grid_params, dtype=np.float64))
        B = gp.shape[0]
        md = self.model_desc(0, 0, 1, 0, 0, 0)
        stride = self.param_stride(md)
        rows = np.zeros((B, stride))
        rows[:, 3] = 1.0
        rows[:, 6 : 6 + self.P] = gp
        with torch.cuda.device(self.dev):
            P = to_dev(rows, self.dev)
            mu = empty((B, self.m), self.dev)
            cov = empty((B, self.m, self.m), self.dev)
            info = empty((B,), self.dev, torch.int32)
            ws = self._work(md, B)
            rc = self.lib.sf_emulator_query_batch(
                self.ctx, C.byref(md), B, ptr(P), ptr(mu), ptr(cov), ptr(info), ptr(ws), ws.numel(),
                stream_ptr(self.dev),
            )
            _lib.check(rc, "sf_emulator_query_batch")
            return mu.cpu().numpy(), cov.cpu().numpy(), info.cpu().numpy()

    def emulator_query_joint(self, grid_params):
        """grid_params: (B, P).  Joint conditional over the B points, component-major: mu (B*m,), cov (B*m, B*m)."""
        torch = _torch()
        gp = np.atleast_2d(np.asarray(grid_params, dtype=np.float64))
        B = gp.shape[0]
        md = self.model_desc(0, 0, 1, 0, 0, 0)
        stride = self.param_stride(md)
        rows = np.zeros((B, stride))
        rows[:, 3] = 1.0
        rows[:, 6 : 6 + self.P] = gp
        with torch.cuda.device(self.dev):
            P = to_dev(rows, self.dev)
            mu = empty((B * self.m,), self.dev)
            cov = empty((B * self.m, B * self.m), self.dev)
            info = empty((B,), self.dev, torch.int32)
            ws = self._work(md, B)
            rc = self.lib.sf_emulator_joint_batch(
                self.ctx, C.byref(md), B, ptr(P), ptr(mu), ptr(cov), ptr(info), ptr(ws), ws.numel(),
                stream_ptr(self.dev),
            )
            _lib.check(rc, "sf_emulator_joint_batch")
            return mu.cpu().numpy(), cov.cpu().numpy(), info.cpu().numpy()
