"""Throughput of the drop-in Python API (SpectrumModel.log_likelihood_batch: host-side packing, H2D/D2H copies and
the device path) at BASELINE config 2, for both solvers.  python tools/bench_python_api.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from starfish_amd import Spectrum, synth
from starfish_amd.emulator import Emulator
from starfish_amd.models import SpectrumModel
o = synth.make_order(N=4096)
emu = Emulator(o["grid_points"], o["param_names"], o["emu_wl"], o["weights"], o["eigenspectra"], o["w_hat"], o["flux_mean"], o["flux_std"], o["factors"]); emu._trained = True
data = Spectrum(o["wave"], o["flux"], sigmas=o["sigma"])
c = dict(synth.centre_params(o)); gp = c.pop("grid_params")
for solver in ("auto", "dense"):
    m = SpectrumModel(emu, data, grid_params=gp, solver=solver, **c)
    P = synth.walker_ball(o, B=128, seed=1)
    m.log_likelihood_batch(P); m.log_likelihood_batch(P)
    n = 20 if solver == "auto" else 3
    t0 = time.perf_counter()
    for _ in range(n): ll = m.log_likelihood_batch(P)
    dt = (time.perf_counter() - t0) / n
    print(f"solver={solver}: SpectrumModel.log_likelihood_batch(128 walkers): {dt*1e3:.2f} ms per call -> {128/dt:.0f} evals/s (python API, host-side packing and copies included)")

# the multi-order model (cfg-3 shape) through the same API, dense solver: host packing of 25 orders, plan set-up, copies
if "--echelle" in sys.argv:
    orders = synth.make_echelle(25, 3000)
    em = synth.build_echelle(orders)
    Ps = synth.shared_ball(orders[0], B=64, seed=1)
    em.log_likelihood_batch(Ps)
    t0 = time.perf_counter()
    for _ in range(3): tot = em.log_likelihood_batch(Ps)
    dt = (time.perf_counter() - t0) / 3
    print(f"EchelleModel.log_likelihood_batch(25 orders x 3000 px x 64 walkers), dense: {dt*1e3:.1f} ms per call -> {1600/dt:.0f} order-evals/s")
