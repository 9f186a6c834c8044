"""Repeat sf_potrf_batch on the dataflow sequence and count calls that took suspiciously long or failed (tuning aid):
    python tools/stress_potrf.py [N] [B] [calls] [sequence]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch

from starfish_amd import _device as D
from starfish_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lib = _lib.require_gpu()
assert lib.sf_debug_cholesky_sequence(int(sys.argv[4]) if len(sys.argv) > 4 else 4) == 0
dev = D.device_of()
lda = N + 16
g = torch.Generator(device=dev).manual_seed(0)
base = torch.empty((N, lda), dtype=torch.float64, device=dev)
base.normal_(generator=g)
base[:, :N] = (base[:, :N] + base[:, :N].T) * 0.01
base[:, :N] += torch.eye(N, dtype=torch.float64, device=dev) * 4.0
A = torch.empty((B, N, lda), dtype=torch.float64, device=dev)
info = torch.empty((B,), dtype=torch.int32, device=dev)
ws = D.workspace(lib.sf_potrf_workspace_bytes(N, B), dev)
s = D.stream_ptr(dev)
ref = None
slow = bad = flagged = 0
times = []
for it in range(calls):
    A.copy_(base.unsqueeze(0).expand(B, N, lda))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(lib.sf_potrf_batch(D.ptr(A), N, lda, N * lda, B, D.ptr(info), D.ptr(ws), ws.numel(), s))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    times.append(dt)
    L = torch.tril(A[:, :, :N])
    if ref is None:
        ref = L.clone()  # (matrices of queues of different sizes take different split factors: compare call to call)
        assert (ref[0] @ ref[0].T - base[:, :N]).abs().max().item() < 1e-11
    if int(info.abs().max()) != 0:
        flagged += 1
    elif not torch.equal(L, ref):
        bad += 1
    if dt > 0.5:
        slow += 1
times.sort()
print(f"N={N} B={B}: {calls} calls, median {times[len(times) // 2] * 1e3:.2f} ms, max {times[-1] * 1e3:.1f} ms, slow (> 0.5 s) {slow}, flagged {flagged}, results differing from the first call {bad}")
