#!/usr/bin/env python
"""
bench.py -- benchmark of the log-likelihood hot path (BASELINE.json).

    python bench.py [--gpus N --steps K --warmup W] [--config cfg2|cfg3|cfg5] [--scaling weak|strong]

Default = the headline: BASELINE cfg 2, log-likelihood evals/sec on a synthetic 4096-pixel order, batch = 128
walkers, fp64.  One STEP = one pass of the whole hot path (emulator query, transform chain, fused covariance
fill, batched Cholesky, solve) over one batch of (walker x order) units, parameters and static order data
already resident in HBM.  The model is built through the product API (starfish_amd.Emulator / Spectrum /
SpectrumModel / EchelleModel, including the model's own init-time resample); bench.py only skips the per-call
host packing by keeping the packed parameter rows on the device.

  cfg2   one order N = 4096, m = 8, B = 128 walkers, all 13 parameters thawed            (headline)
  cfg3   25 orders x N = 3000, B = 64 shared walkers = 1600 units through sf_loglike_multi_batch
         (cfg 4 = the same units sharded order-major over ranks: --gpus N --scaling strong)
  cfg5   one order N = 16384, B = 32

Multi-GPU: the units are independent, every rank (one process per GPU) evaluates its own
slice on its own GPU, NO data-path collective; the process group (nccl == RCCL) only carries the barrier and
the max-over-ranks of the timing.  `python bench.py --gpus N` with N > 1 and no RANK in the environment
starts the N ranks itself (re-exec under torch.distributed.run on 127.0.0.1) and fails loudly when fewer than
N GPUs are visible; launched under torch.distributed.run by somebody else it is one of the ranks.
--scaling weak (default): every rank gets a full batch; strong: the batch of the config is split over the
ranks (128/G walkers per GPU).  With N > 1 the line carries BOTH: the headline in the requested mode and the
other one as the sub-object `strong` / `weak`.  `value` = units of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line which also carries
  roofline      -- the dominant kernel (k_chol_panel, the fused fp64 MFMA panel step of the batched Cholesky):
                   algorithmic flops of its launches / their HIP-event time on the launch streams
  cpu_baseline  -- the CPU oracle (numpy/scipy restatement == the reference's algorithm) timed on the host cores
                   over a bounded sample of the same walkers (N = 1 only).  The oracle is imported ONLY there.
  other_configs -- (default run only: N = 1, cfg 2, no overrides) short legs of cfg 3 and cfg 5
  strong_scaling_proxy -- (same condition) cfg 2 at the per-rank batches of 2 / 4 / 8 ranks on this one GPU
"""
import argparse
import ctypes as C
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X dense FP64 matrix peak (datasheet)
HBM_PEAK_GBS = 8000.0
N_EIG = 8

CONFIGS = {
    "cfg2": dict(npix=4096, batch=128, orders=1, label="cfg2: synthetic single order"),
    "cfg3": dict(npix=3000, batch=64, orders=25, label="cfg3: multi-order model"),
    "cfg5": dict(npix=16384, batch=32, orders=1, label="cfg5: long-order stress"),
}
SHARE_GPU_ENV = "SF_BENCH_RANKS_SHARE_GPU"  # test hook: every rank on device 0, gloo instead of RCCL


# ------------------------------------------------------------------------------------------- CPU baseline
def _cpu_pool_worker(job):
    """cpu_baseline, pool mode: `evals` walkers through the CPU oracle in a fresh process with few BLAS threads
    (the reference's recommended way to use many cores: one process per chain / order, docs/intro.rst:71-73).
    Returns the values and the wall-clock interval of the timed evaluations."""
    order_args, plist, blas_threads = job
    sys.path.insert(0, ROOT)
    from threadpoolctl import threadpool_limits

    from oracle import sf_oracle as O

    with threadpool_limits(limits=blas_threads):
        oo = O.OracleOrder(*order_args)
        O.log_likelihood(oo, plist[0])  # first call pays the one-off set-up of the order (not timed)
        t0 = time.time()
        vals = [O.log_likelihood(oo, p) for p in plist]
        return vals, t0, time.time()


def _cpu_pool_baseline(oo_args, plist, npix, evals=3, blas_threads=4, timeout=600):
    """`evals` walkers per process on ALL host cores (cpu_count / blas_threads processes, memory permitting).
    The rate is MEASURED: all timed evaluations / (last end - first start) on the common wall clock."""
    import multiprocessing as mp

    try:
        import psutil

        avail_gb = psutil.virtual_memory().available / 2**30
    except Exception:
        avail_gb = 64.0
    ncpu = os.cpu_count() or 1
    per_proc_gb = max(0.5, 12 * 8 * npix * npix / 2**30)  # ~a dozen N x N temporaries in the reference's algorithm
    procs = int(max(1, min(len(plist), ncpu // max(1, blas_threads), avail_gb * 0.6 // per_proc_gb)))
    if procs < 2:
        return None
    ctx = mp.get_context("spawn")  # never fork a process that holds a HIP context
    # process p evaluates walkers p, p + procs, ... (wrapping around the batch)
    idx = [[(p + k * procs) % len(plist) for k in range(evals)] for p in range(procs)]
    jobs = [(oo_args, [plist[i] for i in ii], blas_threads) for ii in idx]
    from concurrent.futures import ProcessPoolExecutor

    try:
        with ProcessPoolExecutor(max_workers=procs, mp_context=ctx) as ex:
            t0 = time.perf_counter()
            res = list(ex.map(_cpu_pool_worker, jobs, timeout=timeout))
            wall = time.perf_counter() - t0
    except Exception:
        return None
    span = max(r[2] for r in res) - min(r[1] for r in res)
    flat_idx = [i for ii in idx for i in ii]
    flat_val = [v for r in res for v in r[0]]
    return dict(rate=procs * evals / span, procs=procs, blas_threads=blas_threads, evals=evals, idx=flat_idx,
                vals=flat_val, span=span, wall=wall)


def cpu_baseline(order, plist, lnl_gpu, args):
    """The oracle (kind 'port') over walkers of the same batch: single process with the default BLAS threads, and a
    pool over all cores; the better rate is `value`."""
    import numpy as np

    from oracle import sf_oracle as O  # test infrastructure: only the checker / CPU baseline may import it

    oo_args = (order["wave"], order["flux"], order["sigma"], order["emu_wl"], order["eigenspectra"],
               order["flux_mean"], order["flux_std"], order["grid_points"], order["w_hat"])
    oo = O.OracleOrder(*oo_args)
    k = min(args.cpu_sample, len(plist))
    O.log_likelihood(oo, plist[0])  # warm the BLAS threads
    tc = time.perf_counter()
    want = np.array([O.log_likelihood(oo, p) for p in plist[:k]])
    tcpu = time.perf_counter() - tc
    rel = np.abs(lnl_gpu[:k] - want) / np.abs(want)
    assert rel.max() < 1e-8, rel
    try:
        from threadpoolctl import threadpool_info

        nthreads = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        nthreads = os.cpu_count()
    out = {
        "value": k / tcpu, "unit": "evals/s", "cores": int(nthreads), "kind": "port",
        "sample": f"first {k} walkers of the same batch (order 0) through oracle/sf_oracle.py (numpy/scipy, default "
        f"BLAS threads; host has {os.cpu_count()} logical CPUs)",
        "max_rel_dlnl_vs_gpu": float(rel.max()),
    }
    pool = None if args.no_cpu_pool else _cpu_pool_baseline(oo_args, plist, len(order["wave"]), evals=args.cpu_pool_evals)
    if pool is not None:
        vals = np.array(pool["vals"])
        relp = np.abs(lnl_gpu[pool["idx"]] - vals) / np.abs(vals)
        assert relp.max() < 1e-8, relp
        procs, bt, ev = pool["procs"], pool["blas_threads"], pool["evals"]
        out["single_process"] = {"value": k / tcpu, "cores": int(nthreads)}
        out["process_pool"] = {
            "value": pool["rate"], "processes": procs, "blas_threads_per_process": bt, "cores": procs * bt,
            "evals_per_process": ev, "timed_span_s": pool["span"], "wall_s_incl_startup": pool["wall"],
            "note": "measured aggregate: processes x evals_per_process timed evaluations / (last end - first start)",
        }
        if pool["rate"] > k / tcpu:
            out.update(
                value=pool["rate"], cores=procs * bt,
                sample=f"{procs * ev} evaluations of walkers of the same batch (order 0), {ev} per process ({procs} "
                f"processes x {bt} BLAS threads = all {os.cpu_count()} logical CPUs) through oracle/sf_oracle.py, measured "
                f"aggregate over {pool['span']:.1f} s; single process with {int(nthreads)} BLAS threads: {k / tcpu:.2f} "
                f"evals/s over {k} walkers",
            )
    return out


# ------------------------------------------------------------------------------------------- launcher
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks ourselves, one per GPU,
    rendezvous on 127.0.0.1; rank 0's JSON line is the only thing on stdout."""
    import torch

    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    share = os.environ.get(SHARE_GPU_ENV) == "1"
    if ndev < 1 or (ndev < n and not share):
        sys.stderr.write(f"bench.py: --gpus {n} needs {n} visible GPUs, found {ndev} (one process per GPU; there is no "
                         f"CPU fallback and ranks do not share a GPU unless {SHARE_GPU_ENV}=1 is set for testing)\n")
        sys.exit(2)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    sys.exit(subprocess.call(cmd, env=env))


# ------------------------------------------------------------------------------------------- workloads
class Workload:
    """One config on this rank: the product models, the packed parameter rows on the device, `step()` = one
    enqueue of the whole hot path over this rank's units."""

    def __init__(self, cfg, rank, world, scaling, grid="loguniform", device=None):
        import numpy as np
        import torch

        from starfish_amd import _device as D
        from starfish_amd import parallel, synth

        self.N, self.B, self.n_orders = N, B, n_orders = cfg["npix"], cfg["batch"], cfg["orders"]
        self.units_full = B * n_orders  # units of one full batch of the config
        self.scaling, self.world = scaling, world
        seed = 1 + (rank if scaling == "weak" else 0)
        if n_orders == 1:
            order = synth.make_order(N=N)
            if grid == "perturbed":
                order = synth.perturb_grid(order)
            self.model = model = synth.build_model(order, device=device)
            # weak: every rank its own B walkers (different seeds); strong: the config's B walkers split over the ranks
            P_all = synth.walker_ball(order, B=B, seed=seed)
            lo, hi = (0, B) if scaling == "weak" else parallel.shard_range(B, rank, world)
            self.P = P = P_all[lo:hi]
            self.dev, self.md, self.rows = dev, md, rows = model._pack(P if len(P) else P_all[:1], update_caches=False)
            self.lib, self.nf, self.device = dev.lib, dev.nf, dev.dev
            self.n_local = n_local = hi - lo
            self.P_dev = P_dev = D.to_dev(rows, dev.dev)
            self.lnl = lnl = D.empty((max(n_local, 1),), dev.dev)
            self.info = info = D.empty((max(n_local, 1),), dev.dev, torch.int32)

            def step():
                if n_local:
                    dev.loglike_device(md, P_dev, lnl[:n_local], info[:n_local])

            def results():
                return lnl[:n_local].cpu().numpy(), info[:n_local].cpu().numpy()

            self.plist = [synth.vector_to_oracle_params(p) for p in P]
            self.order0 = order
        else:
            orders = synth.make_echelle(n_orders, N)
            if grid == "perturbed":
                orders = [synth.perturb_grid(o) for o in orders]
            self.model = em = synth.build_echelle(orders, device=device)
            P_all = synth.shared_ball(orders[0], B=B, seed=seed)
            # order-major unit list (order o, walker w) -> this rank's contiguous slice keeps whole orders resident
            lo, hi = (0, self.units_full) if scaling == "weak" else parallel.shard_range(self.units_full, rank, world)
            segs_dev, segs_rows, md = [], [], None
            for o, wlo, whi in parallel.order_major_slices(n_orders, B, lo, hi):
                d_o, md, rows = em.orders[o]._pack(P_all[wlo:whi], update_caches=False)
                segs_dev.append(d_o)
                segs_rows.append(rows)
            self.n_local = hi - lo
            self.plan = plan = D.MultiPlan(segs_dev, md, segs_rows) if segs_dev else None
            d0 = em.orders[0]._device()
            self.lib, self.nf, self.device = d0.lib, d0.nf, d0.dev
            self.dev = None

            def step():
                if plan is not None:
                    plan.enqueue()

            def results():
                if plan is None:
                    return np.zeros(0), np.zeros(0, dtype=np.int32)
                outs = plan.collect()
                return np.concatenate([o["lnl"] for o in outs]), np.concatenate([o["info"] for o in outs])

            # CPU baseline sample: the walkers of order 0 (if this rank owns any)
            first_rows = min(B, hi) - lo if lo < B else 0
            self.plist = [synth.shared_to_oracle_params(orders[0], p) for p in P_all[:max(first_rows, 0)]]
            self.order0 = orders[0]
        self.step, self.results = step, results

    @property
    def units_total(self):
        return self.units_full * self.world if self.scaling == "weak" else self.units_full

    @property
    def flops_eval(self):
        N = self.N
        return N**3 / 3 + 2 * N_EIG * N**2 + N**2

    def release(self):
        """Give the workspaces (17 - 116 GB) back before the next leg is built."""
        import gc

        import torch

        if self.n_orders == 1:
            self.dev.release_workspace()
        else:
            self.plan = None
            for m in self.model.orders:
                m._device().release_workspace()
        self.step = self.results = self.model = self.dev = None
        gc.collect()
        torch.cuda.empty_cache()


def timed_leg(w, steps, warmup, prof_steps, comm, clock=None, _retried=False):
    """W untimed steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; max over ranks.
    Per-launch HIP events (`roofline`) are recorded during the last `prof_steps` timed steps.  `clock`: optional
    (tensor, stream) -- one wave on its own stream samples the shader clock against the 100 MHz wall clock while
    the timed steps run."""
    import numpy as np
    import torch

    from starfish_amd import _device as D

    lib = w.lib
    for _ in range(warmup):
        w.step()
    torch.cuda.synchronize()
    lib.sf_profile_read(None, None, None, None)
    comm.barrier()
    torch.cuda.synchronize()
    if clock is not None:
        lib.sf_debug_clock_probe(D.ptr(clock[0]), 4_000_000, C.c_void_p(clock[1].cuda_stream))
    prof_steps = max(1, min(prof_steps, steps))
    t0 = time.perf_counter()
    for i in range(steps):
        if i == steps - prof_steps:
            lib.sf_profile_enable(1)  # the event records cost ~1 % of a step (measured)
        w.step()
    torch.cuda.synchronize()
    comm.barrier()
    dt = time.perf_counter() - t0
    lib.sf_profile_enable(0)
    dt_max = comm.max(dt)
    ms = (C.c_double * 6)()
    gflops, glaunch, gcalls = C.c_double(), C.c_long(), C.c_long()
    lib.sf_profile_read(ms, C.byref(gflops), C.byref(glaunch), C.byref(gcalls))
    lnl_host, info_host = w.results()
    if comm.max(float((info_host == -5).any())) > 0 and not _retried:  # (agreed over the ranks: the re-run has barriers)
        # SF_INFO_INTERNAL: a bounded wait inside the persistent-kernel Cholesky timed out and the launch was aborted (the
        # kernel needs the GPU to itself: processes sharing a device can starve each other's resident workgroups).  What the
        # product API does (starfish_amd/_device.py): say so, switch this process to the launch sequences, run again.
        sys.stderr.write("bench.py: persistent-kernel Cholesky aborted a launch (info = -5): disabled for this process, leg re-run\n")
        lib.sf_persistent_potrf(0)
        return timed_leg(w, steps, warmup, prof_steps, comm, clock, _retried=True)
    assert (info_host == 0).all(), info_host
    assert np.isfinite(lnl_host).all()
    clock_mhz = None
    if clock is not None:
        ticks, wall = clock[0].cpu().tolist()
        clock_mhz = 100.0 * ticks / wall if wall else None
        if clock_mhz is not None and not (200.0 < clock_mhz < 4000.0):
            clock_mhz = None  # (a probe wave that was context-switched -- ranks sharing a GPU in the tests -- reads nonsense)
    gemm_s = ms[2] * 1e-3
    achieved = gflops.value / gemm_s / 1e12 if gemm_s > 0 else 0.0
    return dict(dt_max=dt_max, steps=steps, warmup=warmup, prof_steps=prof_steps, ms=list(ms), gflops=gflops.value,
                launches=int(glaunch.value), achieved=achieved, clock_mhz=clock_mhz, lnl=lnl_host,
                value=w.units_total * steps / dt_max, ms_per_step=dt_max / steps * 1e3)


def short_summary(w, t):
    """Sub-object of a secondary leg: value, ms_per_step, roofline.frac."""
    return {
        "value": t["value"], "unit": "evals/s", "ms_per_step": t["ms_per_step"], "steps": t["steps"], "warmup": t["warmup"],
        "units_per_gpu": w.n_local, "global_batch": w.units_total,
        "whole_path_frac_of_mfma_peak": t["value"] * w.flops_eval / 1e12 / (FP64_MFMA_PEAK_TFLOPS * w.world),
        "roofline": {"bound": "mfma", "achieved": t["achieved"], "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": t["achieved"] / FP64_MFMA_PEAK_TFLOPS, "profiled_steps": t["prof_steps"]},
        "stage_ms_per_step": {
            k: v / t["prof_steps"]
            for k, v in zip(["transforms", "fill", "panel_union", "potrf_stage", "solve", "panel_launches_sum"], t["ms"])
        },
    }


def code_id():
    """sha1 (16 hex digits) over the sources that determine what is measured: the HIP / C-ABI sources and this file.
    tools/summarize_profile.py stamps the same id into the PMC summaries, so `roofline.traffic` can say whether the
    counters were collected on THIS code (the snapshot on the GPU box has no .git)."""
    import hashlib

    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "starfish_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "starfish_amd", "csrc", "*.cpp"))
                   + glob.glob(os.path.join(ROOT, "starfish_amd", "csrc", "*.h")) + [os.path.join(ROOT, "include", "starfish_amd.h")])
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def fill_unaligned_leg(npix=3000, batch=128, steps=5, warmup=2):
    """The same write-only fill on cfg 3's order size, N = 3000 with row stride N: every other row starts 64 bytes into a
    128-byte line.  Reported next to the row-padded variant (ld = 3008: what a caller that may choose its stride should do)."""
    import torch

    from starfish_amd import _device as D
    from starfish_amd import synth

    order = synth.make_order(N=npix)
    model = synth.build_model(order)
    dev, md, rows = model._pack(synth.walker_ball(order, B=batch, seed=1), update_caches=False)
    lib = dev.lib
    P_dev = D.to_dev(rows, dev.dev)
    info = D.empty((batch,), dev.dev, torch.int32)
    out = {}
    for ld in (npix, -(-npix // 16) * 16):
        cov = torch.empty((batch * npix * ld,), dtype=torch.float64, device=dev.dev)
        for _ in range(warmup):
            dev.cov_fill_device(md, P_dev, cov, ld, npix * ld, lower_only=False, add_jitter=True, info=info)
        torch.cuda.synchronize()
        lib.sf_profile_read(None, None, None, None)
        lib.sf_profile_enable(1)
        for _ in range(steps):
            dev.cov_fill_device(md, P_dev, cov, ld, npix * ld, lower_only=False, add_jitter=True, info=info)
        torch.cuda.synchronize()
        lib.sf_profile_enable(0)
        ms = (C.c_double * 6)()
        lib.sf_profile_read(ms, None, None, None)
        assert (info.cpu().numpy() == 0).all()
        nbytes = 8.0 * npix * npix * batch
        gbs = nbytes / (ms[1] / steps * 1e-3) / 1e9
        out["ld_n" if ld == npix else "ld_padded"] = {"ld": ld, "avg_launch_ms": ms[1] / steps, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS}
        del cov
        torch.cuda.empty_cache()
    dev.release_workspace()
    return {"N": npix, "batch": batch, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "achieved": out["ld_n"]["achieved"], "frac": out["ld_n"]["frac"], "ld_n": out["ld_n"], "ld_padded": out["ld_padded"],
            "note": "sf_cov_fill_batch, full dense C of N = 3000 rows: row stride N (rows alternate between 0 and 64 bytes into a "
            "128-byte line) and row stride 3008 (every row line-aligned); algorithmic bytes 8 N^2 B"}


def fill_leg(w, steps=5, warmup=2):
    """the fill stage alone is HBM-WRITE bound.  sf_cov_fill_batch, full dense matrices (both triangles,
    sigma^2 + K_global + K_local + rank-m term on MFMA + jitter), row stride N: 8 N^2 B algorithmic bytes per launch, all
    written, nothing read back.  Timed with HIP events around the fill launches on their stream (the library's profile
    scopes: the transform chain that precedes the fill is a separate scope).  Next to it the plain streaming write of
    the same array (sf_debug_stream_write): the write rate this box sustains."""
    import torch

    from starfish_amd import _device as D

    dev, md, lib, B, N = w.dev, w.md, w.lib, w.n_local, w.N
    cov = torch.empty((B, N, N), dtype=torch.float64, device=dev.dev)
    info = D.empty((B,), dev.dev, torch.int32)
    nbytes = 8.0 * N * N * B
    for _ in range(warmup):
        dev.cov_fill_device(md, w.P_dev, cov, N, N * N, lower_only=False, add_jitter=True, info=info)
    torch.cuda.synchronize()
    lib.sf_profile_read(None, None, None, None)
    lib.sf_profile_enable(1)
    for _ in range(steps):
        dev.cov_fill_device(md, w.P_dev, cov, N, N * N, lower_only=False, add_jitter=True, info=info)
    torch.cuda.synchronize()
    lib.sf_profile_enable(0)
    ms = (C.c_double * 6)()
    lib.sf_profile_read(ms, None, None, None)
    assert (info.cpu().numpy() == 0).all()
    fill_ms = ms[1] / steps
    # streaming-write probe over the same array
    s = D.stream_ptr(dev.dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        lib.sf_debug_stream_write(D.ptr(cov), cov.numel(), 0.0, s)
    e0.record()
    for _ in range(steps):
        lib.sf_debug_stream_write(D.ptr(cov), cov.numel(), 0.0, s)
    e1.record()
    torch.cuda.synchronize()
    write_ms = e0.elapsed_time(e1) / steps
    del cov
    torch.cuda.empty_cache()
    gbs = nbytes / (fill_ms * 1e-3) / 1e9
    probe = nbytes / (write_ms * 1e-3) / 1e9
    traffic, src, same = pmc_lookup("fill", "k_fill_dense", total=True)
    return {
        "kernel": "k_fill_dense_plain + k_fill_dense_band (side by side on two streams) via sf_cov_fill_batch: full dense C "
        "(both triangles, jitter on), row stride N",
        "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
        "stream_write_probe_gbs": probe, "frac_of_stream_write_probe": gbs / probe if probe > 0 else None,
        "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": fill_ms, "transforms_ms_before_it": ms[0] / steps,
        "launches": steps, "traffic": traffic, "traffic_source": src, "traffic_is_this_code": same,
        "note": "write-only pass: algorithmic bytes = 8 N^2 B; matrices per second = 1e3 B / avg_launch_ms",
        "matrices_per_s": B / (fill_ms * 1e-3),
    }


def sampler_leg(w, steps=10, warmup=2):
    """What an emcee-style loop actually issues (reference caller: examples/single.ipynb:458-466, 528-546):
    a stretch-move ensemble of `w.B` walkers advances by TWO dependent half-ensemble calls per step, each through the drop-in
    front-end SpectrumModel.log_likelihood_batch(P, priors) -- host-side packing of the proposals, priors, upload, the hot
    path on B / 2 walkers, download -- plus the sampler's own host arithmetic.  Reported: evaluations per second of the
    loop, the host time per step outside the device calls, and the ratio to the headline rate (one call of B walkers)."""
    import numpy as np
    import scipy.stats as st
    import torch

    from starfish_amd import samplers, synth

    model, dev = w.model, w.dev
    labels = model.labels
    c = dict(zip(labels, synth.centre_vector(w.order0)))
    priors = {"T": st.uniform(c["T"] - 100, 200), "logg": st.uniform(c["logg"] - 0.5, 1.0), "Z": st.uniform(c["Z"] - 0.5, 1.0),
              "vsini": st.uniform(0, 500), "vz": st.norm(c["vz"], 50.0), "global_cov:log_amp": st.norm(c["global_cov:log_amp"], 5.0),
              "global_cov:log_ls": st.uniform(0, 10)}
    t_dev = [0.0, 0]
    inner = dev.loglike

    def timed_loglike(*a, **k):  # (wall time of the device call as the front-end sees it: enqueue + synchronise + copies)
        t0 = time.perf_counter()
        try:
            return inner(*a, **k)
        finally:
            t_dev[0] += time.perf_counter() - t0
            t_dev[1] += 1

    dev.loglike = timed_loglike
    try:
        nw = w.B
        sampler = samplers.EnsembleSampler(nw, len(labels), lambda P: model.log_likelihood_batch(P, priors), seed=3)
        x = synth.walker_ball(w.order0, B=nw, seed=1)
        x, _ = sampler.run_mcmc(x, warmup)
        torch.cuda.synchronize()
        t_dev[:] = [0.0, 0]
        t0 = time.perf_counter()
        x, lp = sampler.run_mcmc(x, steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        dev.loglike = inner
    evals = (steps + 1) * nw  # (run_mcmc evaluates its start first: one call of all walkers, inside the timed region)
    calls = t_dev[1]
    return {"walkers": nw, "steps": steps, "evals_per_s": evals / dt, "ms_per_step": dt / steps * 1e3,
            "device_calls": calls, "rows_per_call": evals / max(1, calls),
            "device_call_ms_per_step": t_dev[0] / steps * 1e3, "host_ms_per_step": (dt - t_dev[0]) / steps * 1e3,
            "acceptance_fraction": float(np.mean(sampler.acceptance_fraction)), "finite_fraction": float(np.mean(np.isfinite(lp))),
            "note": "starfish_amd.samplers.EnsembleSampler (Goodman-Weare stretch move, two half-ensembles per step) over "
            "SpectrumModel.log_likelihood_batch with seven scipy.stats priors; the first run_mcmc call of the timed region also "
            "evaluates its start (one call of all walkers): counted in evals and in the time"}


def train_leg(w, iterations=25):
    """SpectrumModel.train (spectrum_model.py:635-696): the reference's serial Nelder-Mead loop (B = 1 launches) against the
    batched simplex (starfish_amd/_neldermead.py), same iterations -- wall clock of the whole call, host logic included."""
    import scipy.stats as st

    model = w.model
    x0 = model.get_param_vector().copy()
    out = {}
    model.log_likelihood()
    # (scipy's default simplex enlarges every coordinate by 5 %: T leaves the emulator grid, where the objective raises -- in
    # the reference too; the prior keeps such vertices at -inf without a device call)
    c = dict(zip(model.labels, x0))
    priors = {"T": st.uniform(6000, 200), "vsini": st.uniform(0, 500)}
    for key, kw in (("serial", dict(batch_simplex=False)), ("batched", {})):
        model.set_param_vector(x0)
        t0 = time.perf_counter()
        s = model.train(priors, options=dict(maxiter=iterations), **kw)
        out[key] = {"ms": (time.perf_counter() - t0) * 1e3, "nit": int(s.nit), "nfev": int(s.nfev),
                    "device_calls": int(getattr(s, "nbatches", s.nfev))}
    model.set_param_vector(x0)
    out["speedup"] = out["serial"]["ms"] / out["batched"]["ms"]
    out["note"] = (f"{iterations} Nelder-Mead iterations from the centre parameters (13 thawed), initial simplex included; same "
                   "decisions in both (tests/test_gpu_train.py)")
    return out


def device_note():
    """Name and compute units of rank 0's GPU (boxes of the pool differ by a few per cent at the small batches)."""
    try:
        import torch

        pr = torch.cuda.get_device_properties(torch.cuda.current_device())
        return {"name": pr.name, "compute_units": int(pr.multi_processor_count), "memory_GB": round(pr.total_memory / 2**30, 1)}
    except Exception as e:  # (never fatal: a note)
        return {"error": str(e)[:80]}


def pmc_lookup(tag, kernel_key, total=False, per_step=False):
    """(hbm bytes per launch, source file, collected-on-this-code?) of the latest profiles/r0?_*<tag>*_pmc_summary.json;
    total: summed over every kernel whose name starts with the key (one launch of each per step); per_step: the bytes of
    all launches of one step (hbm_GB_per_step of the panel-kernel aggregate)."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r0[0-9]_*{tag}*_pmc_summary.json")))[-1:]:
        with open(f) as fh:
            summ = json.load(fh)
        field = "hbm_GB_per_step" if per_step else "hbm_bytes_per_launch"
        hits = [val[field] * (1e9 if per_step else 1.0) for key, val in summ.items()
                if isinstance(val, dict) and key.lstrip("_").startswith(kernel_key) and field in val]
        if hits:
            cid = summ.get("_code_id")
            return (sum(hits) if total else hits[0]), os.path.relpath(f, ROOT) + (f" @code {cid}" if cid else ""), (cid == code_id()) if cid else None
    return None, None, None


class Comm:
    """Barrier + max-over-ranks of a scalar: the only things the process group is used for."""

    def __init__(self, dist, device, group=None, dev_index=None):
        self.dist, self.device, self.group, self.dev_index = dist, device, group, dev_index

    def barrier(self):
        if self.dist is None:
            return
        if self.group is not None:  # nccl / RCCL group: the barrier is an all-reduce on this rank's own device
            self.dist.barrier(group=self.group, device_ids=[self.dev_index])
        else:
            self.dist.barrier()

    def sum(self, x):
        """After the timed region: a checksum over the ranks' results (host gather of one double per rank)."""
        if self.dist is None:
            return float(x)
        import torch

        t = torch.tensor([x], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return float(t.item())

    def max(self, x):
        if self.dist is None:
            return float(x)
        import torch

        t = torch.tensor([x], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)  # timing only: the data path has no collective
        return float(t.item())


def make_process_group(world, dev_index, share):
    """(dist, Comm, note).  The default group is gloo (rendezvous on MASTER_ADDR, carries the agreement below and is
    the fallback); the barrier / timing-max group is nccl == RCCL over xGMI, created on top of it and tried with one
    all-reduce.  If RCCL cannot be brought up on ANY rank (agreed with a MIN over gloo) every rank falls back to gloo and
    the JSON line says so in `process_group` -- the data path has no collective either way."""
    import datetime

    import torch
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=900))
    assert dist.get_world_size() == world
    cpu = torch.device("cpu")
    if share:  # test hook: RCCL refuses two ranks on one device
        return dist, Comm(dist, cpu), "gloo (test hook: ranks share device 0)"
    ok, why, grp = 1, "", None
    cuda = torch.device("cuda", dev_index)
    try:
        if os.environ.get("SF_BENCH_FAIL_NCCL") == "1":  # test hook: exercise the fallback
            raise RuntimeError("SF_BENCH_FAIL_NCCL=1")
        grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=300), device_id=cuda)
        t = torch.ones(1, dtype=torch.float64, device=cuda)
        dist.all_reduce(t, group=grp)
        torch.cuda.synchronize()
        assert int(t.item()) == world, t.item()
    except Exception as e:  # noqa: BLE001 -- whatever RCCL raises: report it, do not die
        ok, why = 0, f"{type(e).__name__}: {str(e).splitlines()[0][:200] if str(e) else ''}"
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return dist, Comm(dist, cuda, grp, dev_index), "nccl (RCCL)"
    sys.stderr.write(f"bench.py: rank {os.environ.get('RANK')}: RCCL group unavailable ({why or 'another rank failed'}); using gloo\n")
    return dist, Comm(dist, cpu), "gloo (fallback: the nccl/RCCL group could not be created" + (f": {why}" if why else " on another rank") + ")"


class ErrorLine:
    """Whatever happens, rank 0 prints ONE parsable JSON line.  An exception on rank 0 prints {"error": ...}; a failure
    elsewhere makes the launcher SIGTERM the surviving ranks -- rank 0 may be blocked inside a collective then, where a
    Python signal handler would never run, so the signal's wake-up byte is read by a watchdog THREAD which prints the
    line (with what the failed ranks left in their error files) and exits."""

    def __init__(self, args, rank, world):
        import signal
        import tempfile
        import threading

        self.args, self.rank, self.world, self.printed, self.terminated = args, rank, world, False, False
        self.lock = threading.Lock()
        tag = os.environ.get("MASTER_PORT") or str(os.getppid())
        self.dir = os.path.join(tempfile.gettempdir(), f"sf_bench_errors_{tag}")
        if world > 1 and rank == 0:
            try:
                os.makedirs(self.dir, exist_ok=True)
                for f in glob.glob(os.path.join(self.dir, "rank*.err")):
                    os.remove(f)
                self.rd, wr = socket.socketpair()
                wr.setblocking(False)
                signal.signal(signal.SIGTERM, lambda *a: None)
                signal.siginterrupt(signal.SIGTERM, False)  # SA_RESTART: the signal must not fail a HIP ioctl with EINTR
                signal.set_wakeup_fd(wr.fileno(), warn_on_full_buffer=False)
                self._wr = wr
                threading.Thread(target=self._watch, daemon=True).start()
            except Exception:  # pragma: no cover -- never let the safety net break the bench
                pass

    def _watch(self):
        import signal

        while True:
            b = self.rd.recv(16)
            if not b:
                return
            if signal.SIGTERM in b:
                self.terminated = True
                time.sleep(0.5)  # let the failing rank finish writing its error file
                self.emit("terminated by the launcher (SIGTERM): another rank failed" + self._others())
                os._exit(143)

    def _others(self):
        out = []
        for f in sorted(glob.glob(os.path.join(self.dir, "rank*.err"))):
            try:
                with open(f) as fh:
                    out.append(f"{os.path.basename(f)[:-4]}: {fh.read().strip()[:400]}")
            except OSError:
                pass
        return (" -- " + " | ".join(out)) if out else ""

    def emit(self, msg):
        with self.lock:
            if self.printed or self.rank != 0:
                return
            self.printed = True
            cfg = CONFIGS[self.args.config]
            print(json.dumps({
                "metric": f"log-likelihood evals/sec, {self.args.npix or cfg['npix']}-pixel order, batch={self.args.batch or cfg['batch']} walkers",
                "value": None, "unit": "evals/s", "n_gpus": self.world, "steps": self.args.steps, "warmup": self.args.warmup,
                "ms_per_step": None, "higher_is_better": True, "scaling": self.args.scaling, "vs_baseline": None,
                "dtype": "f64", "data": "synthetic", "config": {"workload": cfg["label"]}, "error": msg,
            }), flush=True)

    def result(self, out):
        with self.lock:
            if not self.printed:
                self.printed = True
                print(json.dumps(out), flush=True)

    def failed(self, exc):
        import traceback

        msg = f"rank {self.rank}: {type(exc).__name__}: {exc}"
        sys.stderr.write(traceback.format_exc())
        if self.rank == 0:
            if self.world > 1:  # an error caused by the launcher's SIGTERM: the watchdog's line names the real culprit
                time.sleep(0.3)
                if self.terminated:
                    time.sleep(5.0)  # (the watchdog prints and exits the process)
            self.emit(msg)
        else:
            try:
                os.makedirs(self.dir, exist_ok=True)
                with open(os.path.join(self.dir, f"rank{self.rank}.err"), "w") as fh:
                    fh.write(msg)
            except OSError:
                pass


def structured_leg(w, args, comm, lnl_host, custom):
    """Secondary figure (single-order configs): the structure-exploiting solver (band + rank-m Woodbury, SURVEY.md 8
    f-4) on the SAME walkers.  It is not the headline `value`: BASELINE's metric is the dense path."""
    import numpy as np
    import torch

    from starfish_amd import _device as D

    dev, md, rows, model, lib, n_local, N = w.dev, w.md, w.rows, w.model, w.lib, w.n_local, w.N
    hw = int(dev.halfwidth_bound(md, rows).max())
    if not (0 <= hw <= dev.banded_max_halfwidth()):
        return None
    lnl_b = D.empty((n_local,), dev.dev)
    info_b = D.empty((n_local,), dev.dev, torch.int32)
    ksteps = max(args.steps, 10)
    for _ in range(2):
        dev.loglike_banded_device(md, w.P_dev, hw, lnl_b, info_b)
    torch.cuda.synchronize()
    lib.sf_profile_read(None, None, None, None)
    lib.sf_profile_enable(1)
    comm.barrier()
    torch.cuda.synchronize()
    tb = time.perf_counter()
    for _ in range(ksteps):
        dev.loglike_banded_device(md, w.P_dev, hw, lnl_b, info_b)
    torch.cuda.synchronize()
    comm.barrier()
    dtb = comm.max(time.perf_counter() - tb)
    lib.sf_profile_enable(0)
    msb = (C.c_double * 6)()
    lib.sf_profile_read(msb, None, None, None)
    lb = lnl_b.cpu().numpy()
    assert (info_b.cpu().numpy() == 0).all()
    rel_b = float(np.max(np.abs(lb - lnl_host) / np.abs(lnl_host)))
    assert rel_b < 1e-9, rel_b
    # roofline of the sweep kernel: the band is read once from HBM; MFMA floor from the per-column block count
    n16 = (N + 15) // 16 * 16
    band_bytes = n_local * n16 * ((hw + 2) & ~1) * 8.0
    sweep_s = msb[3] / ksteps * 1e-3
    nbr = (hw + 15) // 16 + 1
    mfma_per_col = nbr * (nbr + 1) / 2 + 2 * nbr + 2 * (1 + 8) + 24  # window pairs + solves + rhs rows + 16x16 potrf/inverse
    floor_s = mfma_per_col * 64 / 4 * (n16 / 16) / 2.4e9 * max(1.0, n_local / 256.0)
    structured = {
        "value": n_local * w.world * ksteps / dtb, "unit": "evals/s", "ms_per_step": dtb / ksteps * 1e3,
        "steps": ksteps, "band_halfwidth_px": hw, "max_rel_dlnl_vs_dense_path": rel_b,
        "stage_ms_per_step": {
            "transforms": msb[0] / ksteps, "band_fill": msb[1] / ksteps,
            "band_cholesky_forms": msb[3] / ksteps, "woodbury_finish": msb[4] / ksteps,
        },
        "roofline": {
            "kernel": "k_band_forms (LDS-window banded Cholesky + forward substitutions, one workgroup per "
            "matrix or per half matrix)",
            "bound": "latency (sequential 16-column chain per matrix); HBM and MFMA floors for reference",
            "hbm": {"bytes_per_step": band_bytes, "achieved": band_bytes / sweep_s / 1e9 if sweep_s > 0 else None,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": band_bytes / sweep_s / 1e9 / HBM_PEAK_GBS if sweep_s > 0 else None},
            "mfma_floor_ms": floor_s * 1e3, "sweep_ms": sweep_s * 1e3,
            "frac_of_mfma_floor": floor_s / sweep_s if sweep_s > 0 else None,
        },
        "note": "sf_loglike_banded_batch: C = band + Y^T Y never formed; same lnL to rounding; O(N W^2) flops, "
        "so the dense MFMA roofline above does not apply to it",
    }
    # the same walkers with a wider global kernel (ls = 30 km/s: half-width 24 ls / dv = 361 px, beyond the LDS
    # window): the bordered-band factorisation on the dense path's panel kernel (sf_launch_potrf_band)
    labels = list(model.labels)
    if "global_cov:log_ls" in labels and not custom:
        Pw = np.array(w.P, dtype=float, copy=True)
        Pw[:, labels.index("global_cov:log_ls")] = np.log(30.0)
        _, md_w, rows_w = model._pack(Pw, update_caches=False)
        hw_w = int(dev.halfwidth_bound(md_w, rows_w).max())
        if dev.banded_window_halfwidth() < hw_w <= dev.banded_max_halfwidth():
            Pw_dev = D.to_dev(rows_w, dev.dev)
            dense_w = D.empty((n_local,), dev.dev)
            dev.loglike_device(md_w, Pw_dev, dense_w, info_b)
            for _ in range(2):
                dev.loglike_banded_device(md_w, Pw_dev, hw_w, lnl_b, info_b)
            torch.cuda.synchronize()
            tw = time.perf_counter()
            for _ in range(5):
                dev.loglike_banded_device(md_w, Pw_dev, hw_w, lnl_b, info_b)
            torch.cuda.synchronize()
            dtw = (time.perf_counter() - tw) / 5
            lw, dw = lnl_b.cpu().numpy(), dense_w.cpu().numpy()
            assert (info_b.cpu().numpy() == 0).all()
            rel_w = float(np.max(np.abs(lw - dw) / np.abs(dw)))
            assert rel_w < 1e-9, rel_w
            structured["wide_band"] = {
                "global_ls_kms": 30.0, "band_halfwidth_px": hw_w, "ms_per_step": dtw * 1e3,
                "value": n_local / dtw, "unit": "evals/s per GPU", "max_rel_dlnl_vs_dense_path": rel_w,
                "kernel": "bordered band matrix on k_diag_lds + k_chol_panel, K loops limited to the band",
            }
    return structured


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2")
    ap.add_argument("--npix", type=int, default=None, help="override the config's pixels per order")
    ap.add_argument("--batch", type=int, default=None, help="override the config's walkers")
    ap.add_argument("--orders", type=int, default=None, help="override the config's number of orders")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--grid", choices=["loguniform", "perturbed"], default="loguniform",
                    help="perturbed: non log-uniform wavelengths (K_global evaluated per entry)")
    ap.add_argument("--profile-steps", type=int, default=1,
                    help="timed steps (the last ones) during which per-launch HIP events are recorded for `roofline`")
    ap.add_argument("--cpu-sample", type=int, default=8, help="walkers timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-pool-evals", type=int, default=3, help="cpu_baseline pool: timed evaluations per process")
    ap.add_argument("--no-structured", action="store_true", help="skip the banded-solver secondary figure")
    ap.add_argument("--no-cpu-pool", action="store_true", help="cpu_baseline: single-process mode only")
    ap.add_argument("--fill-only", action="store_true",
                    help="run only the HBM-write leg (sf_cov_fill_batch, full dense) and print its object: for rocprofv3 passes")
    ap.add_argument("--single-scaling", action="store_true",
                    help="N > 1: only the requested scaling mode, no sub-object for the other one (the 8-rank rehearsal on one "
                    "GPU: eight full cfg-3 batches would not fit one device)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip other_configs (cfg 3 / cfg 5) and strong_scaling_proxy of the default run")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    in_group = "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ
    if args.gpus > 1 and not in_group:
        launch_ranks(args.gpus, sys.argv[1:])  # does not return

    rank = int(os.environ.get("RANK", "0")) if in_group else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if in_group else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if in_group else 1
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)\n")
        sys.exit(2)
    line = ErrorLine(args, rank, world)
    try:
        run(args, in_group, rank, local_rank, world, line)
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001 -- rank 0 still prints one parsable line; the launcher sees a failure
        line.failed(e)
        sys.stdout.flush()
        os._exit(1)  # (not sys.exit: a wedged process group must not keep the interpreter in its destructors)


def run(args, in_group, rank, local_rank, world, line):
    import numpy as np
    import torch

    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU fallback)"
    share = os.environ.get(SHARE_GPU_ENV) == "1"
    ndev = torch.cuda.device_count()
    if world > 1 and not share and local_rank >= ndev:
        sys.stderr.write(f"bench.py: rank {rank} has no GPU of its own ({ndev} visible, one process per GPU)\n")
        sys.exit(2)
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    # (SF_BENCH_FORCE_GROUP=1: take the process-group branch with ONE rank too -- RCCL accepts a single rank; the -m gpu
    # test uses it to run init, barrier, all-reduce and the clock-probe order of the N > 1 path on the one-GPU box)
    use_dist = world > 1 or (in_group and os.environ.get("SF_BENCH_FORCE_GROUP") == "1")
    if os.environ.get("SF_BENCH_FAIL_RANK") == str(rank):  # test hook: this rank dies before the process group exists
        raise RuntimeError("SF_BENCH_FAIL_RANK test hook")

    cfg = dict(CONFIGS[args.config])
    custom = []
    for key in ("npix", "batch", "orders"):
        v = getattr(args, key)
        if v is not None and v != cfg[key]:
            cfg[key] = v
            custom.append(f"{key}={v}")

    # ---- build the model(s) through the product API and pack this rank's parameter rows
    w = Workload(cfg, rank, world, args.scaling, args.grid)
    N, B, n_orders, device, lib = w.N, w.B, w.n_orders, w.device, w.lib
    if share and world > 1:
        # test hook only: several ranks on ONE device.  The persistent-kernel Cholesky spins on counters written by its own
        # other workgroups: it needs them all resident, i.e. the GPU to itself -- one process per GPU, as the metric says.
        # Eight processes oversubscribing a device deadlock each other's launches (measured: aborted by the 4-s bound).
        lib.sf_persistent_potrf(0)
    clk = torch.zeros(2, dtype=torch.int64, device=device)
    clk_stream = torch.cuda.Stream(device=device)
    clock = None if os.environ.get("SF_BENCH_NO_CLOCK") else (clk, clk_stream)
    clock_note = "sampled by one wave on its own stream while the timed steps ran"
    clock_mhz = None
    dist = None
    if use_dist:
        # With RCCL initialised the spinning probe kernel serialises with the main stream (measured: +40 ms on the
        # timed region), so under a process group the sustained clock is sampled BEFORE the group is created, over
        # untimed steps of the same workload on this rank's GPU.
        if clock is not None and w.n_local:
            pre = timed_leg(w, 2, max(1, args.warmup), 1, Comm(None, device), clock)
            clock_mhz = pre["clock_mhz"]
            clock_note = "sampled on rank 0 over 2 untimed steps of the same workload before the process group was created"
            clock = None
        dist, comm, pg_note = make_process_group(world, dev_index, share)
    else:
        comm, pg_note = Comm(None, device), None

    if args.fill_only:
        assert n_orders == 1 and not use_dist, "--fill-only: single-order configs on one GPU"
        print(json.dumps({"fill": fill_leg(w, steps=args.steps, warmup=args.warmup)}), flush=True)
        return
    t = timed_leg(w, args.steps, args.warmup, args.profile_steps, comm, clock if not use_dist else None)
    if not use_dist:
        clock_mhz = t["clock_mhz"]
    lnl_host = t["lnl"]
    lnl_checksum = comm.sum(float(np.sum(lnl_host)))  # sum of lnL over ALL units of all ranks (strong: = the 1-rank value)
    structured = None
    if n_orders == 1 and not args.no_structured and w.n_local:
        structured = structured_leg(w, args, comm, lnl_host, custom)

    # ---- N > 1: the other scaling mode as a sub-object (SURVEY.md 8e split: units of ONE batch over the ranks)
    other_mode, other = None, None
    if use_dist and not args.single_scaling:
        other_mode = "strong" if args.scaling == "weak" else "weak"
        plist0, order0 = w.plist, w.order0
        w.release()
        w2 = Workload(cfg, rank, world, other_mode, args.grid)
        t2 = timed_leg(w2, args.steps, max(2, args.warmup), args.profile_steps, comm)  # (a new batch size: >= 2 untimed steps)
        other = short_summary(w2, t2)
        other["scaling"] = other_mode
        other["n_gpus"] = world
        w2.release()
    default_run = (world == 1 and args.config == "cfg2" and not custom and args.grid == "loguniform"
                   and not args.no_extra_legs)
    extra, proxy, proxy5, fill, sampler, train = None, None, None, None, None, None
    if default_run:
        plist0, order0 = w.plist, w.order0
        base_ms_per_eval = t["ms_per_step"] / w.n_local
        sampler = sampler_leg(w)
        sampler["efficiency_vs_headline"] = sampler["evals_per_s"] / t["value"]
        train = train_leg(w)
        fill = fill_leg(w)
        w.release()
        fill["unaligned"] = fill_unaligned_leg()
        # strong-scaling proxy on the one GPU: the per-rank batch of 2 / 4 / 8 ranks (128/G walkers)
        proxy = [{"ranks_equivalent": 1, "batch": B, "value": t["value"], "ms_per_step": t["ms_per_step"],
                  "per_eval_efficiency_vs_full_batch": 1.0}]
        for g in (2, 4, 8):
            c = dict(cfg, batch=B // g)
            wp = Workload(c, 0, 1, "weak")
            tp = timed_leg(wp, 5, 2, 1, comm)
            proxy.append({"ranks_equivalent": g, "batch": B // g, "value": tp["value"], "ms_per_step": tp["ms_per_step"],
                          "per_eval_efficiency_vs_full_batch": base_ms_per_eval / (tp["ms_per_step"] / wp.n_local),
                          "panel_frac_of_peak": tp["achieved"] / FP64_MFMA_PEAK_TFLOPS})
            wp.release()
        extra = {}
        for name, steps_x in (("cfg3", 3), ("cfg5", 3)):
            wx = Workload(CONFIGS[name], 0, 1, "weak")
            tx = timed_leg(wx, steps_x, 1, 1, comm)
            extra[name] = short_summary(wx, tx)
            extra[name]["workload"] = CONFIGS[name]["label"] + f": {wx.n_orders} order(s) x N_pix={wx.N}, batch={wx.B}"
            extra[name]["ceiling_evals_per_s"] = FP64_MFMA_PEAK_TFLOPS * 1e12 / wx.flops_eval
            wx.release()
        # ... and of cfg 5 (N = 16384, 32 walkers: 16 / 8 / 4 per GPU) -- long chunks are what docs/intro.rst:73 warns about
        c5 = CONFIGS["cfg5"]
        ms5 = extra["cfg5"]["ms_per_step"] / c5["batch"]
        proxy5 = [{"ranks_equivalent": 1, "batch": c5["batch"], "value": extra["cfg5"]["value"],
                   "ms_per_step": extra["cfg5"]["ms_per_step"], "per_eval_efficiency_vs_full_batch": 1.0}]
        for g in (2, 4, 8):
            wp = Workload(dict(c5, batch=c5["batch"] // g), 0, 1, "weak")
            tp = timed_leg(wp, 2, 1, 1, comm)
            proxy5.append({"ranks_equivalent": g, "batch": wp.B, "value": tp["value"], "ms_per_step": tp["ms_per_step"],
                           "per_eval_efficiency_vs_full_batch": ms5 / (tp["ms_per_step"] / wp.n_local),
                           "panel_frac_of_peak": tp["achieved"] / FP64_MFMA_PEAK_TFLOPS})
            wp.release()
    elif not use_dist or args.single_scaling:
        plist0, order0 = w.plist, w.order0

    if rank == 0:
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (FETCH_SIZE x2 as the
        # micro-arch guide prescribes for gfx950, + WRITE_SIZE) summarised under profiles/ by
        # tools/summarize_profile.py; bench.py cannot collect counters itself.
        traffic, traffic_src, traffic_same = (None, None, None)
        if not custom and args.grid == "loguniform":
            traffic, traffic_src, traffic_same = pmc_lookup(args.config, "k_chol_panel_all", per_step=True)
        ms, prof_steps, achieved = t["ms"], t["prof_steps"], t["achieved"]
        label = cfg["label"] + (" [custom: " + ", ".join(custom) + "]" if custom else "")
        per = " per GPU" if args.scaling == "weak" else " in total"
        if n_orders == 1:
            workload = (f"{label} N_pix={N}, 8 eigenspectra, M=27, N_f={w.nf}, 1 global + 1 local kernel, all 13 "
                        f"parameters thawed, batch={B} walkers" + per)
            metric = f"log-likelihood evals/sec, {N}-pixel order, batch={B} walkers"
        else:
            workload = (f"{label}: {n_orders} echelle orders x N_pix={N}, 8 eigenspectra, M=27, N_f={w.nf}, shared walkers "
                        f"batch={B} (10 thawed parameters, every order its own local kernel) = {w.units_full} (order x walker) "
                        "units" + per + ", one sf_loglike_multi_batch pass")
            metric = f"single-order log-likelihood evals/sec (order x walker units), {n_orders} orders x {N} pixels, batch={B} walkers"
        if args.grid == "perturbed":
            workload += "; NON log-uniform wavelength grid (K_global per entry)"
        out = {
            "metric": metric,
            "value": t["value"],
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": t["ms_per_step"],
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "global_batch": w.units_total,
                "units_per_gpu": w.n_local,
                "parallelism": f"(order x walker) units sharded x{world}, one process per GPU, no data-path collective"
                + (f" (process group: {pg_note}, barrier + timing max only)" if use_dist else ""),
                "device": device_note(),
            },
            "process_group": pg_note,
            "lnl_checksum": lnl_checksum,
            "persistent_potrf_enabled": bool(lib.sf_persistent_potrf(-1)),
            "whole_path_tflops": t["value"] * w.flops_eval / 1e12,
            "whole_path_frac_of_mfma_peak": t["value"] * w.flops_eval / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world),
            "roofline": None,  # (filled in below: the flat scalars first -- the driver keeps only the first scalar keys)
            "stage_ms_per_step": {
                k: v / prof_steps
                for k, v in zip(["transforms", "fill", "panel_union", "potrf_stage", "solve", "panel_launches_sum"], ms)
            },
            "potrf_stage_tflops": w.n_local * prof_steps * (N**3 / 3) / (ms[3] * 1e-3) / 1e12 if ms[3] > 0 else None,
            "structured_solver": structured,
        }
        # ---- roofline: flat scalars FIRST (the driver's record keeps the leading scalar keys and drops nested objects).
        # `frac` is the WHOLE-PATH figure: algorithmic flops of the path (N^3/3 + 2 m N^2 + N^2 per eval) / step time as
        # the driver clocks it / the datasheet peak; the panel kernels' own figure (their algorithmic flops / the union of
        # their HIP-event intervals) is `panel_frac`.
        whole_tf = t["value"] * w.flops_eval / 1e12 / world
        alg_bytes_step = 16.0 * N * N * w.n_local  # C written once + read once
        # (key order: the driver's record keeps the first ~24 scalars.  Schema since round 5: `frac` = whole path; the panel
        # kernels' figure, `frac` of rounds 1-4, is `panel_frac`.)
        roof = {"bound": "mfma", "achieved": whole_tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": whole_tf / FP64_MFMA_PEAK_TFLOPS, "panel_frac": achieved / FP64_MFMA_PEAK_TFLOPS}
        if sampler is not None:
            roof["sampler_efficiency"] = sampler["efficiency_vs_headline"]
            roof["sampler_host_ms_per_step"] = sampler["host_ms_per_step"]
        if train is not None:
            roof["train_batched_speedup"] = train["speedup"]
        if fill is not None:
            roof["fill_frac"] = fill["frac"]
            if fill.get("unaligned"):
                roof["fill_unaligned_frac"] = fill["unaligned"]["frac"]
        if proxy is not None:
            for row in proxy[1:]:
                roof[f"b{row['batch']}_efficiency"] = row["per_eval_efficiency_vs_full_batch"]
        if extra is not None:
            roof["cfg3_frac"] = extra["cfg3"]["whole_path_frac_of_mfma_peak"]
            roof["cfg5_frac"] = extra["cfg5"]["whole_path_frac_of_mfma_peak"]
            roof["cfg3_panel_frac"] = extra["cfg3"]["roofline"]["frac"]
            roof["cfg5_panel_frac"] = extra["cfg5"]["roofline"]["frac"]
        if proxy5 is not None:
            for row in proxy5[1:]:
                roof[f"cfg5_b{row['batch']}_efficiency"] = row["per_eval_efficiency_vs_full_batch"]
        roof["traffic"] = traffic
        roof["traffic_ratio"] = traffic / alg_bytes_step if traffic else None  # counter bytes per step / 16 N^2 B
        roof["schema"] = "frac = whole path (rounds 1-4: frac = panel kernels, now panel_frac)"
        roof["whole_path_frac"] = whole_tf / FP64_MFMA_PEAK_TFLOPS
        roof["panel_achieved"] = achieved
        if fill is not None:
            roof["fill_gbs"] = fill["achieved"]
            if fill.get("unaligned"):
                roof["fill_unaligned_gbs"] = fill["unaligned"]["achieved"]
        roof["traffic_is_this_code"] = traffic_same
        roof["sustained_clock_mhz"] = clock_mhz or None
        roof["peak_at_sustained_clock"] = FP64_MFMA_PEAK_TFLOPS * clock_mhz / 2400.0 if clock_mhz else None
        roof["profiled_steps"] = prof_steps
        roof["launches"] = t["launches"]
        roof["avg_launch_ms"] = ms[5] / max(1, t["launches"])
        roof["achieved_by_sum_of_launch_durations"] = t["gflops"] / (ms[5] * 1e-3) / 1e12 if ms[5] > 0 else None
        roof["algorithmic_flops_per_launch"] = t["gflops"] / max(1, t["launches"])
        roof["algorithmic_bytes_per_step"] = alg_bytes_step
        roof["code_id"] = code_id()
        roof["traffic_source"] = traffic_src
        roof["sustained_clock_note"] = clock_note if clock_mhz else None
        roof["kernel"] = ("k_chol_panel_w / k_chol_panel / k_potrf_dataflow (fused v_mfma_f64_16x16x4_f64 panel steps of the batched "
                          "Cholesky: long-K update + triangular solves + diagonal-tile update; panel pairs when batch x slabs >= 3400, "
                          "128-column panels for their chain and for medium batches, ONE persistent dataflow launch while batch x "
                          "panels <= 2048)")
        roof["note"] = ("`frac` / `achieved`: whole path -- algorithmic flops of every stage / the timed step; `panel_frac` / "
                        "`panel_achieved`: rank 0's panel launches, which run on several streams (lookahead chain, slab groups) and "
                        "overlap: their algorithmic flops / the UNION of the launch intervals (HIP events on the launch streams, common "
                        "origin, recorded during the last `profiled_steps` of the timed steps); avg_launch_ms is the plain mean launch "
                        "duration (what rocprofv3 --stats reports); `traffic` = HBM bytes per step from the rocprofv3 --pmc passes "
                        "under profiles/ (FETCH_SIZE x 2 + WRITE_SIZE), traffic_ratio = traffic / algorithmic_bytes_per_step")
        if fill is not None:
            roof["fill"] = fill  # "the fill stage alone is HBM-write-bound ... report both"
        out["roofline"] = roof
        if other is not None:
            out[other_mode] = other
        if proxy is not None:
            out["strong_scaling_proxy"] = {
                "note": "1-GPU proxy of the strong split (SURVEY.md 8e: 128/G walkers per GPU): cfg 2 at the per-rank batch "
                "of G ranks, 5 timed steps each", "rows": proxy,
                "cfg5_rows": proxy5, "cfg5_note": "the same for cfg 5 (N = 16384): 32 / G walkers per GPU, 2 timed steps each"}
        if extra is not None:
            out["other_configs"] = extra
        if sampler is not None:
            out["sampler_step"] = sampler
        if train is not None:
            out["train"] = train
        if world == 1 and args.cpu_sample > 0 and plist0:
            out["cpu_baseline"] = cpu_baseline(order0, plist0, lnl_host, args)
        line.result(out)
    if use_dist:
        comm.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
