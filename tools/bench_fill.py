"""Timing of the dense covariance fill alone (sf_cov_fill_batch, both triangles):  python tools/bench_fill.py [N] [B] [ld]
Prints ms per launch and GB/s for the model with / without structured kernels, next to the streaming-write probe.
Tuning switches (SF_FILL_SPAN, SF_FILL_OLD) need the tuning build: SF_LIB_PATH=starfish_amd/libstarfish_amd_tuning.so."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from starfish_amd import _device as D
from starfish_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ld = int(sys.argv[3]) if len(sys.argv) > 3 else N
o = synth.make_order(N=N)
out = {"N": N, "B": B, "ld": ld, "env": {k: v for k, v in os.environ.items() if k.startswith("SF_FILL")}}
for name, params in (("structured", None), ("rank_m_only", {k: v for k, v in synth.centre_params(o).items() if k not in ("global_cov", "local_cov")})):
    model = synth.build_model(o, params=params)
    P = synth.walker_ball(o, B=B, seed=1)
    if params is not None:
        keep = [i for i, k in enumerate(synth.LABELS) if not (k.startswith("global_cov") or k.startswith("local_cov"))]
        P = P[:, keep]
    dev, md, rows = model._pack(P, update_caches=False)
    lib = dev.lib
    P_dev = D.to_dev(rows, dev.dev)
    cov = torch.empty((B * N * ld,), dtype=torch.float64, device=dev.dev)
    info = D.empty((B,), dev.dev, torch.int32)
    for _ in range(2):
        dev.cov_fill_device(md, P_dev, cov, ld, N * ld, False, True, info)
    torch.cuda.synchronize()
    lib.sf_profile_read(None, None, None, None)
    lib.sf_profile_enable(1)
    K = 5
    for _ in range(K):
        dev.cov_fill_device(md, P_dev, cov, ld, N * ld, False, True, info)
    torch.cuda.synchronize()
    lib.sf_profile_enable(0)
    ms = (C.c_double * 6)()
    lib.sf_profile_read(ms, None, None, None)
    t = ms[1] / K
    out[name] = {"ms": t, "GBs": 8.0 * N * N * B / t / 1e6}
    if name == "structured":
        s = D.stream_ptr(dev.dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lib.sf_debug_stream_write(D.ptr(cov), B * N * N, 0.0, s)
        e0.record()
        for _ in range(K):
            lib.sf_debug_stream_write(D.ptr(cov), B * N * N, 0.0, s)
        e1.record()
        torch.cuda.synchronize()
        tw = e0.elapsed_time(e1) / K
        out["stream_write"] = {"ms": tw, "GBs": 8.0 * N * N * B / tw / 1e6}
    dev.release_workspace()
    del cov
    torch.cuda.empty_cache()
print(json.dumps(out))
