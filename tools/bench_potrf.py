"""Stand-alone timing of sf_potrf_batch on synthetic SPD matrices (kernel tuning aid).
    python tools/bench_potrf.py [N] [B] [reps] [sequence]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch

from starfish_amd import _device as D
from starfish_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lib = _lib.require_gpu()
if len(sys.argv) > 4:  # launch sequence: 0 fused (128-column panels), 1 unfused, 2 wide (panel pairs)
    assert lib.sf_debug_cholesky_sequence(int(sys.argv[4])) == 0
dev = D.device_of()
lda = N + 16
# diagonally dominant random symmetric matrix generated on the device (plumbing only)
g = torch.Generator(device=dev).manual_seed(0)
base = torch.empty((N, lda), dtype=torch.float64, device=dev)
base.normal_(generator=g)
base[:, :N] = (base[:, :N] + base[:, :N].T) * 0.01
base[:, :N] += torch.eye(N, dtype=torch.float64, device=dev) * 4.0
A = torch.empty((B, N, lda), dtype=torch.float64, device=dev)
info = torch.empty((B,), dtype=torch.int32, device=dev)
ws = D.workspace(lib.sf_potrf_workspace_bytes(N, B), dev)
s = D.stream_ptr(dev)
ms = (C.c_double * 6)()
fl, nl, nc = C.c_double(), C.c_long(), C.c_long()
for it in range(reps + 1):
    A.copy_(base.unsqueeze(0).expand(B, N, lda))
    torch.cuda.synchronize()
    lib.sf_profile_read(None, None, None, None)
    lib.sf_profile_enable(1)
    t0 = time.perf_counter()
    _lib.check(lib.sf_potrf_batch(D.ptr(A), N, lda, N * lda, B, D.ptr(info), D.ptr(ws), ws.numel(), s))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.sf_profile_enable(0)
    lib.sf_profile_read(ms, C.byref(fl), C.byref(nl), C.byref(nc))
    if it:
        print(f"N={N} B={B}: potrf {dt*1e3:.2f} ms = {B*N**3/3/dt/1e12:.1f} TF whole;  mfma kernels {ms[2]:.2f} ms "
              f"({fl.value/ms[2]/1e9:.1f} TF algorithmic over {nl.value} launches)")
# sustained shader clock while the factorisation runs (probe wave on its own stream)
probe = torch.zeros(2, dtype=torch.int64, device=dev)
side = torch.cuda.Stream(device=dev)
A.copy_(base.unsqueeze(0).expand(B, N, lda))
torch.cuda.synchronize()
lib.sf_debug_clock_probe(D.ptr(probe), 4_000_000, C.c_void_p(side.cuda_stream))  # 40 ms window
_lib.check(lib.sf_potrf_batch(D.ptr(A), N, lda, N * lda, B, D.ptr(info), D.ptr(ws), ws.numel(), s))
torch.cuda.synchronize()
t, w = probe.cpu().tolist()
print(f"shader clock during potrf: {100.0 * t / w:.0f} MHz  (fp64 MFMA peak at that clock: {78.6 * (t / w) / 24.0:.1f} TFLOP/s)")
lib.sf_debug_clock_probe(D.ptr(probe), 1_000_000, C.c_void_p(side.cuda_stream))
torch.cuda.synchronize()
t, w = probe.cpu().tolist()
print(f"shader clock idle: {100.0 * t / w:.0f} MHz")
assert int(info.abs().max()) == 0
L = torch.tril(A[0, :, :N])
err = (L @ L.T - base[:, :N]).abs().max().item()
print("max |L L^T - A| =", err)
