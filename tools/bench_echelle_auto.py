"""cfg-3 shape through the drop-in EchelleModel API with the structure-exploiting solver per order
(host packing, copies and synchronisations included).  python tools/bench_echelle_auto.py [orders] [npix] [walkers]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from starfish_amd import synth

n_orders = int(sys.argv[1]) if len(sys.argv) > 1 else 25
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
orders = synth.make_echelle(n_orders, N)
em = synth.build_echelle(orders)
P = synth.shared_ball(orders[0], B=B, seed=1)
dense = em.log_likelihood_batch(P)
for m in em.orders:
    m.solver = "auto"
auto = em.log_likelihood_batch(P)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    auto = em.log_likelihood_batch(P)
dt = (time.perf_counter() - t0) / reps
print(f"EchelleModel {n_orders} orders x {N} px x {B} walkers, solver='auto': {dt * 1e3:.1f} ms per call = "
      f"{n_orders * B / dt:.0f} order-evals/s; max rel dlnL vs the dense pass {np.max(np.abs(auto - dense) / np.abs(dense)):.2e}")
