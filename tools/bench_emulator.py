"""Time the emulator query kernels (k_emu_prep, k_emu_z, k_emu_post) at a realistic library size: m = 4, M = 330 (m M = 1320).
    python tools/bench_emulator.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch

from starfish_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for tag, kw in (("M=27 m=8", dict(m=8)), ("M=330 m=4", dict(m=4, grid_axes=synth.BIG_GRID_AXES)), ("M=330 m=8", dict(m=8, grid_axes=synth.BIG_GRID_AXES))):
    o = synth.make_order(N=256, seed=13, **kw)
    t0 = time.perf_counter()
    m = synth.build_model(o)
    dev = m._device()
    t_init = time.perf_counter() - t0
    rng = np.random.default_rng(0)
    lo, hi = np.min(o["grid_points"], 0), np.max(o["grid_points"], 0)
    q = lo + (hi - lo) * rng.uniform(0.05, 0.95, (B, 3))
    dev.emulator_query(q)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dev.emulator_query(q)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{tag}: init (Emulator + context, incl. v11 factor) {t_init*1e3:.1f} ms; emulator_query B={B} (incl. host copies) {dt*1e3:.3f} ms")
