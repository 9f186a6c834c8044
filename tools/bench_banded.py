"""Time the structure-exploiting solver (sf_loglike_banded_batch) at a BASELINE config and compare with
the dense path.  Usage: [SF_BENCH_LS=km/s] python tools/bench_banded.py [npix] [batch] [steps]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from gpu_helpers import device_order, oracle_order, pack_rows
from starfish_amd import _device as D
from starfish_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
order = synth.make_order(N=N)
oo = oracle_order(order)
do = device_order(oo)
P = synth.walker_ball(order, B=B, seed=1)
plist = [synth.vector_to_oracle_params(p) for p in P]
if os.environ.get("SF_BENCH_LS"):  # wider global kernel: half-width = 24 ls / dv pixels
    for p in plist:
        p["global_cov"] = (p["global_cov"][0], float(np.log(float(os.environ["SF_BENCH_LS"]))))
md, rows = pack_rows(do, plist)
hw = int(do.halfwidth_bound(md, rows).max())
P_dev = D.to_dev(rows, do.dev)
lnl = D.empty((B,), do.dev)
info = D.empty((B,), do.dev, torch.int32)
for _ in range(3):
    do.loglike_banded_device(md, P_dev, hw, lnl, info)
torch.cuda.synchronize()
do.lib.sf_profile_read(None, None, None, None)
do.lib.sf_profile_enable(1)
t0 = time.perf_counter()
for _ in range(steps):
    do.loglike_banded_device(md, P_dev, hw, lnl, info)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
do.lib.sf_profile_enable(0)
ms = (C.c_double * 6)()
do.lib.sf_profile_read(ms, None, None, None)
assert (info.cpu().numpy() == 0).all()
band = lnl.cpu().numpy().copy()
print(f"banded: N={N} B={B} halfwidth={hw}: {dt / steps * 1e3:.3f} ms/step -> {B * steps / dt:.0f} evals/s; "
      f"stages ms/step: transforms {ms[0] / steps:.3f} bandfill {ms[1] / steps:.3f} band_forms {ms[3] / steps:.3f} "
      f"woodbury+finish {ms[4] / steps:.3f}")
if os.environ.get("SF_COMPARE_DENSE", "1") != "0" and N <= 8192:
    do.loglike_device(md, P_dev, lnl, info)
    torch.cuda.synchronize()
    dense = lnl.cpu().numpy()
    print("max rel |lnL_banded - lnL_dense| =", float(np.max(np.abs(band - dense) / np.abs(dense))))
