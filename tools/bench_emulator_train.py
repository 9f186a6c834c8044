"""One objective call of Emulator.train at the reference's worked-example size (m = 4, M = 330: a 1320 x 1320
Cholesky of v11 + the solve, Starfish/emulator/emulator.py:484-524,602-619): set_param_vector (host bookkeeping only;
v11 is lazy) + log_likelihood (device: sf_emulator_v11_build + sf_potrf_batch + sf_logdet_sqmah_batch), and the
reference's way -- numpy build of v11 + scipy cho_factor / cho_solve on the host -- for scale.
Round 3, one MI355X box: 0.0 + 2.6 ms per objective call (round 2: 16.8 ms host rebuild of v11 + 3.4 ms with its upload);
host numpy + scipy 15.9 ms.    python tools/bench_emulator_train.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from scipy.linalg import cho_factor, cho_solve

from starfish_amd import synth
from starfish_amd.emulator import Emulator

o = synth.make_order(N=256, m=4, seed=13, grid_axes=synth.BIG_GRID_AXES)
emu = Emulator(o["grid_points"], o["param_names"], o["emu_wl"], o["weights"], o["eigenspectra"], o["w_hat"],
               o["flux_mean"], o["flux_std"], o["factors"])
P = emu.get_param_vector()
emu.log_likelihood()
reps = 10
t0 = time.perf_counter()
for i in range(reps):
    emu.set_param_vector(P + 1e-3 * i)
t_set = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for i in range(reps):
    emu.set_param_vector(P + 1e-3 * i)
    val = emu.log_likelihood()
t_ll = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for i in range(reps):
    emu.set_param_vector(P + 1e-3 * i)  # (invalidates the lazy host v11: the loop below pays its numpy build, like the reference)
    f = cho_factor(emu.v11)
    ref = -(2 * np.sum(np.log(f[0].diagonal())) + emu.w_hat @ cho_solve(f, emu.w_hat)) / 2
t_host = (time.perf_counter() - t0) / reps
print(f"m M = {emu.v11.shape[0]}: set_param_vector {t_set * 1e3:.1f} ms; log_likelihood on the device {t_ll * 1e3:.1f} ms; "
      f"host numpy v11 + scipy cho_factor + cho_solve {t_host * 1e3:.1f} ms; rel diff {abs(val - ref) / abs(ref):.1e}")
