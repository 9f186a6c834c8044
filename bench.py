#!/usr/bin/env python
"""
bench.py -- benchmark of the log-likelihood hot path (BASELINE.json).

    python bench.py [--config cfg2|cfg3|cfg5] [--gpus N --steps K --warmup W] [--scaling weak|strong]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Default = the headline: BASELINE cfg 2, log-likelihood evals/sec on a synthetic 4096-pixel order, batch = 128
walkers, fp64.  One STEP = one pass of the whole hot path (emulator query, transform chain, fused covariance
fill, batched Cholesky, solve) over one batch of (walker x order) units, parameters and static order data
already resident in HBM.  The model is built through the product API (starfish_amd.Emulator / Spectrum /
SpectrumModel / EchelleModel, including the model's own init-time resample); bench.py only skips the per-call
host packing by keeping the packed parameter rows on the device.

  cfg2   one order N = 4096, m = 8, B = 128 walkers, all 13 parameters thawed            (headline)
  cfg3   25 orders x N = 3000, B = 64 shared walkers = 1600 units through sf_loglike_multi_batch
         (cfg 4 = the same units sharded over ranks: --gpus N --scaling strong)
  cfg5   one order N = 16384, B = 32

Multi-GPU: the units are independent, every rank evaluates its own slice on its own GPU, no data-path
collective.  --scaling weak (default): every rank gets a full batch; --scaling strong: the batch of the config
is split over the ranks (128/G walkers per GPU).  `value` = units of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line which also carries
  roofline     -- the dominant kernel (k_chol_panel, the fused fp64 MFMA panel step of the batched Cholesky):
                  algorithmic flops of its launches / their HIP-event time on the launch streams
  cpu_baseline -- the CPU oracle (numpy/scipy restatement == the reference's algorithm) timed on the host cores
                  over a bounded sample of the same walkers (N = 1 only).  The oracle is imported ONLY there.
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X dense FP64 matrix peak (datasheet)
HBM_PEAK_GBS = 8000.0

CONFIGS = {
    "cfg2": dict(npix=4096, batch=128, orders=1, label="cfg2: synthetic single order"),
    "cfg3": dict(npix=3000, batch=64, orders=25, label="cfg3: multi-order model"),
    "cfg5": dict(npix=16384, batch=32, orders=1, label="cfg5: long-order stress"),
}


# ------------------------------------------------------------------------------------------- CPU baseline
def _cpu_pool_worker(job):
    """cpu_baseline, pool mode: one walker through the CPU oracle in a fresh process with few BLAS threads
    (the reference's recommended way to use many cores: one process per chain / order, docs/intro.rst:71-73)."""
    order_args, params, blas_threads = job
    sys.path.insert(0, ROOT)
    from threadpoolctl import threadpool_limits

    from oracle import sf_oracle as O

    with threadpool_limits(limits=blas_threads):
        oo = O.OracleOrder(*order_args)
        O.log_likelihood(oo, params)  # first call pays the one-off set-up of the order (not timed)
        t0 = time.perf_counter()
        val = O.log_likelihood(oo, params)
        return val, time.perf_counter() - t0


def _cpu_pool_baseline(oo_args, plist, npix, blas_threads=4, timeout=300):
    """One walker per process on ALL host cores (cpu_count / blas_threads processes, memory permitting);
    returns (evals/s, processes, threads/process, lnL values, wall) or None."""
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
    jobs = [(oo_args, p, blas_threads) for p in plist[:procs]]
    from concurrent.futures import ProcessPoolExecutor

    try:
        with ProcessPoolExecutor(max_workers=procs, mp_context=ctx) as ex:
            t0 = time.perf_counter()
            res = list(ex.map(_cpu_pool_worker, jobs, timeout=timeout))
            wall = time.perf_counter() - t0
    except Exception:
        return None
    per_eval = max(r[1] for r in res)  # steady state: every process keeps evaluating at its measured rate
    return procs / per_eval, procs, blas_threads, [r[0] for r in res], wall


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
    pool = None if args.no_cpu_pool else _cpu_pool_baseline(oo_args, plist, len(order["wave"]))
    if pool is not None:
        rate, procs, bt, vals, wall = pool
        relp = np.abs(lnl_gpu[: len(vals)] - np.array(vals)) / np.abs(np.array(vals))
        assert relp.max() < 1e-8, relp
        out["single_process"] = {"value": k / tcpu, "cores": int(nthreads)}
        out["process_pool"] = {
            "value": rate, "processes": procs, "blas_threads_per_process": bt, "cores": procs * bt,
            "wall_s_incl_startup": wall,
            "note": "one walker per process, rate = processes / slowest per-eval time (steady state)",
        }
        if rate > k / tcpu:
            out.update(
                value=rate, cores=procs * bt,
                sample=f"{procs} walkers of the same batch (order 0), one per process ({procs} processes x {bt} BLAS "
                f"threads = all {os.cpu_count()} logical CPUs) through oracle/sf_oracle.py; single process with "
                f"{int(nthreads)} BLAS threads: {k / tcpu:.2f} evals/s over {k} walkers",
            )
    return out


# ------------------------------------------------------------------------------------------- workloads
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
    ap.add_argument("--no-structured", action="store_true", help="skip the banded-solver secondary figure")
    ap.add_argument("--no-cpu-pool", action="store_true", help="cpu_baseline: single-process mode only")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched by torch.distributed.run
    if os.environ.get("SF_BENCH_NO_DIST"):
        use_dist = False
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == world

    from starfish_amd import _device as D
    from starfish_amd import parallel, synth

    cfg = dict(CONFIGS[args.config])
    custom = []
    for key in ("npix", "batch", "orders"):
        v = getattr(args, key)
        if v is not None and v != cfg[key]:
            cfg[key] = v
            custom.append(f"{key}={v}")
    N, B, n_orders = cfg["npix"], cfg["batch"], cfg["orders"]
    units_full = B * n_orders  # units of one full batch of the config

    # ---- build the model(s) through the product API and pack this rank's parameter rows
    lib = None
    if n_orders == 1:
        order = synth.make_order(N=N)
        if args.grid == "perturbed":
            order = synth.perturb_grid(order)
        model = synth.build_model(order)
        # weak: every rank its own B walkers (different seeds); strong: the config's B walkers split over the ranks
        P_all = synth.walker_ball(order, B=B, seed=1 + (rank if args.scaling == "weak" else 0))
        lo, hi = (0, B) if args.scaling == "weak" else parallel.shard_range(B, rank, world)
        P = P_all[lo:hi]
        dev, md, rows = model._pack(P, update_caches=False)
        lib = dev.lib
        P_dev = D.to_dev(rows, dev.dev)
        n_local = hi - lo
        lnl = D.empty((max(n_local, 1),), dev.dev)
        info = D.empty((max(n_local, 1),), dev.dev, torch.int32)
        nf = dev.nf
        device = dev.dev

        def step():
            if n_local:
                dev.loglike_device(md, P_dev, lnl[:n_local], info[:n_local])

        def results():
            return lnl[:n_local].cpu().numpy(), info[:n_local].cpu().numpy()

        plist = [synth.vector_to_oracle_params(p) for p in P]
        order0 = order
    else:
        orders = synth.make_echelle(n_orders, N)
        if args.grid == "perturbed":
            orders = [synth.perturb_grid(o) for o in orders]
        em = synth.build_echelle(orders)
        P_all = synth.shared_ball(orders[0], B=B, seed=1 + (rank if args.scaling == "weak" else 0))
        # order-major unit list (order o, walker w) -> this rank's contiguous slice keeps whole orders resident
        lo, hi = (0, units_full) if args.scaling == "weak" else parallel.shard_range(units_full, rank, world)
        segs_dev, segs_rows = [], []
        for o, m in enumerate(em.orders):
            wlo, whi = max(lo, o * B) - o * B, min(hi, (o + 1) * B) - o * B
            if whi <= wlo:
                continue
            d_o, md, rows = m._pack(P_all[wlo:whi], update_caches=False)
            segs_dev.append(d_o)
            segs_rows.append(rows)
        n_local = hi - lo
        plan = D.MultiPlan(segs_dev, md, segs_rows) if segs_dev else None
        lib = em.orders[0]._device().lib
        nf = em.orders[0]._device().nf
        device = em.orders[0]._device().dev

        def step():
            if plan is not None:
                plan.enqueue()

        def results():
            if plan is None:
                return np.zeros(0), np.zeros(0, dtype=np.int32)
            outs = plan.collect()
            return np.concatenate([o["lnl"] for o in outs]), np.concatenate([o["info"] for o in outs])

        # CPU baseline sample: the walkers of order 0
        first_rows = min(B, hi) - lo if lo < B else 0
        plist = [synth.shared_to_oracle_params(orders[0], p) for p in P_all[:max(first_rows, 0)]]
        order0 = orders[0]

    def barrier():
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    lib.sf_profile_read(None, None, None, None)
    barrier()
    torch.cuda.synchronize()
    # one wave on its own stream samples the shader clock against the 100 MHz wall clock while the timed
    # steps run (sustained clock under this load; the datasheet peak assumes 2.4 GHz)
    clk = torch.zeros(2, dtype=torch.int64, device=device)
    clk_stream = torch.cuda.Stream(device=device)
    # (skipped under torch.distributed: with RCCL initialised the spinning probe kernel serialises with the
    # main stream -- measured +40 ms on the timed region)
    if not use_dist and not os.environ.get("SF_BENCH_NO_CLOCK"):
        lib.sf_debug_clock_probe(D.ptr(clk), 4_000_000, C.c_void_p(clk_stream.cuda_stream))
    # HIP events on the launch streams around every panel-kernel launch (and every stage) are recorded during the
    # LAST `--profile-steps` of the timed steps (the event records cost ~1 % of a step, measured)
    prof_steps = max(1, min(args.profile_steps, args.steps))
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - prof_steps:
            lib.sf_profile_enable(1)
        step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    lib.sf_profile_enable(0)

    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # timing only: the data path has no collective
    dt_max = float(t.item())
    units_total = units_full * world if args.scaling == "weak" else units_full

    ms = (C.c_double * 6)()
    gflops, glaunch, gcalls = C.c_double(), C.c_long(), C.c_long()
    lib.sf_profile_read(ms, C.byref(gflops), C.byref(glaunch), C.byref(gcalls))
    lnl_host, info_host = results()
    assert (info_host == 0).all(), info_host
    assert np.isfinite(lnl_host).all()

    # ---- secondary figure (single-order configs): the structure-exploiting solver (band + rank-m Woodbury,
    # SURVEY.md 8 f-4) on the SAME walkers.  It is not the headline `value`: BASELINE's metric is the dense path.
    structured = None
    if n_orders == 1 and not args.no_structured and n_local:
        hw = int(dev.halfwidth_bound(md, rows).max())
        if 0 <= hw <= dev.banded_max_halfwidth():
            lnl_b = D.empty((n_local,), dev.dev)
            info_b = D.empty((n_local,), dev.dev, torch.int32)
            ksteps = max(args.steps, 10)
            for _ in range(2):
                dev.loglike_banded_device(md, P_dev, hw, lnl_b, info_b)
            torch.cuda.synchronize()
            lib.sf_profile_read(None, None, None, None)
            lib.sf_profile_enable(1)
            barrier()
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(ksteps):
                dev.loglike_banded_device(md, P_dev, hw, lnl_b, info_b)
            torch.cuda.synchronize()
            barrier()
            dtb = time.perf_counter() - tb
            lib.sf_profile_enable(0)
            tt = torch.tensor([dtb], dtype=torch.float64, device=dev.dev)
            if use_dist:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtb = float(tt.item())
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
                "value": n_local * world * ksteps / dtb, "unit": "evals/s", "ms_per_step": dtb / ksteps * 1e3,
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
                Pw = np.array(P, dtype=float, copy=True)
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

    if rank == 0:
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (FETCH_SIZE x2 as the
        # micro-arch guide prescribes for gfx950, + WRITE_SIZE) summarised under profiles/ by
        # tools/summarize_profile.py; bench.py cannot collect counters itself.
        traffic, traffic_src = None, None
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r02_*{args.config}*_pmc_summary.json")))[-1:]:
            if not custom and args.grid == "loguniform":
                with open(f) as fh:
                    summ = json.load(fh)
                key = "_k_chol_panel_all" if "_k_chol_panel_all" in summ else None
                if key:
                    traffic = summ[key]["hbm_bytes_per_launch"]
                    traffic_src = os.path.relpath(f, ROOT)
        ticks, wall = clk.cpu().tolist()
        clock_mhz = 100.0 * ticks / wall if wall else 0.0
        gemm_s = ms[2] * 1e-3
        achieved = gflops.value / gemm_s / 1e12 if gemm_s > 0 else 0.0
        flops_eval = N**3 / 3 + 2 * 8 * N**2 + N**2
        label = cfg["label"] + (" [custom: " + ", ".join(custom) + "]" if custom else "")
        if n_orders == 1:
            workload = (f"{label} N_pix={N}, 8 eigenspectra, M=27, N_f={nf}, 1 global + 1 local kernel, all 13 "
                        f"parameters thawed, batch={B} walkers" + (" per GPU" if args.scaling == "weak" else " in total"))
            metric = f"log-likelihood evals/sec, {N}-pixel order, batch={B} walkers"
        else:
            workload = (f"{label}: {n_orders} echelle orders x N_pix={N}, 8 eigenspectra, M=27, N_f={nf}, shared walkers "
                        f"batch={B} (10 thawed parameters, every order its own local kernel) = {units_full} (order x walker) "
                        "units" + (" per GPU" if args.scaling == "weak" else " in total") + ", one sf_loglike_multi_batch pass")
            metric = f"single-order log-likelihood evals/sec (order x walker units), {n_orders} orders x {N} pixels, batch={B} walkers"
        if args.grid == "perturbed":
            workload += "; NON log-uniform wavelength grid (K_global per entry)"
        out = {
            "metric": metric,
            "value": units_total * args.steps / dt_max,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "global_batch": units_total,
                "units_per_gpu": n_local,
                "parallelism": f"(order x walker) units sharded x{world}, no collective",
            },
            "whole_path_tflops": units_total * args.steps * flops_eval / dt_max / 1e12,
            "whole_path_frac_of_mfma_peak": units_total * args.steps * flops_eval / dt_max / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world),
            "roofline": {
                "kernel": "k_chol_panel (fused v_mfma_f64_16x16x4_f64 panel step of the batched Cholesky: long-K update "
                "+ triangular solve + diagonal-tile update)",
                "bound": "mfma",
                "achieved": achieved,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                "sustained_clock_mhz": clock_mhz or None,
                "peak_at_sustained_clock": FP64_MFMA_PEAK_TFLOPS * clock_mhz / 2400.0 if clock_mhz else None,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "note": "the panel launches run on several streams (lookahead chain, slab groups) and overlap: `achieved` "
                "divides by the UNION of the launch intervals (HIP events, common origin, recorded during the last "
                "`profiled_steps` of the timed steps); avg_launch_ms is the plain mean launch duration (what rocprofv3 "
                "--stats reports)",
                "profiled_steps": prof_steps,
                "launches": int(glaunch.value),
                "avg_launch_ms": ms[5] / max(1, glaunch.value),
                "achieved_by_sum_of_launch_durations": gflops.value / (ms[5] * 1e-3) / 1e12 if ms[5] > 0 else None,
                "algorithmic_flops_per_launch": gflops.value / max(1, glaunch.value),
            },
            "stage_ms_per_step": {
                k: v / prof_steps
                for k, v in zip(["transforms", "fill", "panel_union", "potrf_stage", "solve", "panel_launches_sum"], ms)
            },
            "potrf_stage_tflops": n_local * prof_steps * (N**3 / 3) / (ms[3] * 1e-3) / 1e12 if ms[3] > 0 else None,
            "structured_solver": structured,
        }
        if world == 1 and args.cpu_sample > 0 and plist:
            out["cpu_baseline"] = cpu_baseline(order0, plist, lnl_host, args)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
