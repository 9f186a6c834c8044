"""The persistent-kernel ("dataflow") Cholesky is the library's own choice for the batches of a strong split and for every
scalar evaluation.  These tests hold its failure handling to the reference's behaviour at
Starfish/models/spectrum_model.py:400 (cho_factor either factors or raises -- a proposal is never silently rejected):

* a walker whose covariance is not positive definite INSIDE a 16- / 32-matrix dataflow launch: LAPACK's pivot index,
  -inf for that walker only, bit-identical values for the others, LinAlgError from the scalar API;
* a launch that aborts (forced through the tuning build: `make TUNING=1`, SF_DF_FORCE_ABORT) comes back
  SF_INFO_INTERNAL for the whole batch -> the host layer warns, switches the process to the launch sequences and re-runs:
  same values as the fused sequence, bit for bit, for SpectrumModel (batch and scalar), EchelleModel and Emulator;
* a dispenser that never claims a chain task (SF_DF_MISS_CLAIMS): the waits' rescue path (claim inside the wait, run the
  chain task, resume the queued task) alone carries every chain -- same factors bit for bit.
Needs an MI355X: run with -m gpu."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import sf_oracle as O
from starfish_amd import synth

from gpu_helpers import oracle_order

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
TUNING_LIB = os.path.join(ROOT, "starfish_amd", "libstarfish_amd_tuning.so")


def close_lnl(got, want):
    return abs(got - want) <= 1e-8 * abs(want) + 1e-8


@pytest.mark.parametrize("B", [16, 32])
def test_non_positive_definite_walkers_inside_a_dataflow_launch(B):
    """N = 4096, B = 16 / 32: the per-GPU batches of an 8- / 4-way split of cfg 2, on the library's OWN choice of sequence
    (the persistent kernel: the same call with the sequence pinned to it gives the same bits)."""
    from starfish_amd import _lib

    lib = _lib.require_gpu()
    N, bad = 4096, (3, B - 1)
    o = dict(synth.make_order(N=N))
    o["sigma"] = np.zeros(N)  # noise-free order: with log_scale = 18 the rank-m term swamps everything else
    model = synth.build_model(o)
    P = synth.walker_ball(o, B=B, seed=21)
    Pbad = P.copy()
    Pbad[list(bad), 2] = 18.0
    assert lib.sf_persistent_potrf(-1) == 1
    good, info0 = model.log_likelihood_batch(P, return_info=True)
    assert (info0 == 0).all() and np.isfinite(good).all()
    got, info = model.log_likelihood_batch(Pbad, return_info=True)
    try:
        assert lib.sf_debug_cholesky_sequence(4) == 0
        pinned, info_p = model.log_likelihood_batch(Pbad, return_info=True)
        assert lib.sf_debug_cholesky_sequence(0) == 0
        fused = model.log_likelihood_batch(P)
    finally:
        lib.sf_debug_cholesky_sequence(-1)
    np.testing.assert_array_equal(got, pinned)  # the automatic choice IS the dataflow sequence
    np.testing.assert_array_equal(info, info_p)
    assert not np.array_equal(good, fused)  # ... and not the fused one (different summation order: different last bits)
    m = 8
    for b in range(B):
        if b in bad:
            assert got[b] == -np.inf and m < info[b] <= m + 8, (b, got[b], info[b])  # oracle / LAPACK: the 9-th minor
        else:
            assert info[b] == 0 and got[b] == good[b], (b, got[b], good[b])
    oo = oracle_order(o)
    for b in (0, B - 2):
        assert close_lnl(got[b], O.log_likelihood(oo, synth.vector_to_oracle_params(P[b])))
    with pytest.raises(np.linalg.LinAlgError):
        O.log_likelihood(oo, synth.vector_to_oracle_params(Pbad[3]))
    model.set_param_vector(Pbad[3])
    with pytest.raises(np.linalg.LinAlgError, match="leading minor"):
        model.log_likelihood()  # (B = 1: a dataflow launch of one matrix)
    model.set_param_vector(P[3])
    assert model.log_likelihood() == pytest.approx(good[3], rel=1e-9)


def _tuning_lib():
    if not os.path.exists(TUNING_LIB):  # (normally built by __graft_entry__.build() and shipped with the tree)
        subprocess.run(["make", "-C", os.path.join(ROOT, "starfish_amd", "csrc"), "-j8", "TUNING=1"], check=True,
                       capture_output=True)
    return TUNING_LIB


def _run(code, **env):
    e = dict(os.environ, SF_LIB_PATH=_tuning_lib(), PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    e.update(env)
    r = subprocess.run([sys.executable, "-W", "always", "-c", code], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line), r.stderr


_MODEL = r"""
import json, warnings, numpy as np
from starfish_amd import synth, _lib
lib = _lib.require_gpu()
o = synth.make_order(N=1024, m=4, seed=5)
model = synth.build_model(o)
P = synth.walker_ball(o, B=16, seed=3)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    lnl, info = model.log_likelihood_batch(P, return_info=True)
    model.set_param_vector(P[2])
    scalar = model.log_likelihood()
print(json.dumps(dict(lnl=lnl.tolist(), info=info.tolist(), scalar=scalar, enabled=lib.sf_persistent_potrf(-1),
                      warned=[str(x.message) for x in w if issubclass(x.category, RuntimeWarning)])))
"""


def test_an_aborted_dataflow_launch_warns_and_is_rerun_on_the_fused_sequence():
    """Dense path, SpectrumModel: the first call aborts (forced), is re-run after sf_persistent_potrf(0); every later call
    of the process takes a launch sequence.  Values = those of the fused sequence, bit for bit."""
    aborted, _ = _run(_MODEL, SF_DF_FORCE_ABORT="1")
    assert aborted["enabled"] == 0 and len(aborted["warned"]) == 1 and "internal status -5" in aborted["warned"][0]
    assert aborted["info"] == [0] * 16
    # the same process without the forced abort but with the persistent kernel switched off up front
    clean, _ = _run(_MODEL.replace("lib = _lib.require_gpu()", "lib = _lib.require_gpu(); lib.sf_persistent_potrf(0)"))
    assert clean["warned"] == [] and clean["enabled"] == 0
    assert aborted["lnl"] == clean["lnl"] and aborted["scalar"] == clean["scalar"]
    # ... and the untouched default (dataflow) agrees to rounding
    dflt, _ = _run(_MODEL)
    assert dflt["warned"] == [] and dflt["enabled"] == 1
    np.testing.assert_allclose(aborted["lnl"], dflt["lnl"], rtol=1e-11)


_SCALAR_FIRST = r"""
import json, warnings, numpy as np
from starfish_amd import synth, _lib
lib = _lib.require_gpu()
o = synth.make_order(N=512, m=4, seed=5)
model = synth.build_model(o)
orders = synth.make_echelle(n_orders=3, N=320, m=4, seed0=40)
ech = synth.build_echelle(orders)
Pe = synth.shared_ball(orders[0], B=4, seed=2)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    cnt = lambda: sum("internal status -5" in str(x.message) for x in w)
    first = model.log_likelihood()         # B = 1: the persistent kernel -> aborted -> re-run
    n1 = cnt()
    lib.sf_persistent_potrf(1)             # back on: the emulator's own factorisation aborts next
    e = model.emulator.log_likelihood()
    n2 = cnt()
    lib.sf_persistent_potrf(1)             # ... and the multi-order call (12 units, one sf_loglike_multi_batch)
    multi = ech.log_likelihood_batch(Pe)
    n3 = cnt()
print(json.dumps(dict(first=first, emu=e, multi=multi.tolist(), n=[n1, n2, n3], enabled=lib.sf_persistent_potrf(-1),
                      msgs=[str(x.message) for x in w])))
"""


def test_scalar_emulator_and_multi_order_likelihoods_recover_from_an_aborted_launch():
    """Every host entry that factors on the dense path: SpectrumModel.log_likelihood (B = 1), Emulator.log_likelihood
    (sf_potrf_batch on v11) and EchelleModel.log_likelihood_batch (MultiPlan) each meet one forced abort, warn once and
    return the value of the launch sequences."""
    got, _ = _run(_SCALAR_FIRST, SF_DF_FORCE_ABORT="1")
    want, _ = _run(_SCALAR_FIRST)
    assert got["n"] == [1, 2, 3], got["msgs"]
    assert want["n"] == [0, 0, 0] and want["enabled"] == 1 and got["enabled"] == 0
    assert got["first"] == pytest.approx(want["first"], rel=1e-11)
    assert got["emu"] == pytest.approx(want["emu"], rel=1e-11)
    np.testing.assert_allclose(got["multi"], want["multi"], rtol=1e-11)
    assert np.isfinite(got["multi"]).all()


_POTRF = r"""
import json, sys, numpy as np, torch
from starfish_amd import _device as D, _lib
lib = _lib.require_gpu()
assert lib.sf_debug_cholesky_sequence(4) == 0
out = {}
for N, B in ((2048, 4), (1024, 16), (3008, 3)):
    dev = D.device_of(); lda = N + 16
    g = torch.Generator(device=dev).manual_seed(N + B)
    base = torch.empty((N, lda), dtype=torch.float64, device=dev).normal_(generator=g)
    base[:, :N] = (base[:, :N] + base[:, :N].T) * 0.01 + torch.eye(N, dtype=torch.float64, device=dev) * 4.0
    A = base.unsqueeze(0).repeat(B, 1, 1)
    for b in range(B):
        A[b, :, :N] += torch.eye(N, dtype=torch.float64, device=dev) * (0.01 * b)
    info = torch.empty((B,), dtype=torch.int32, device=dev)
    ws = D.workspace(lib.sf_potrf_workspace_bytes(N, B), dev)
    _lib.check(lib.sf_potrf_batch(D.ptr(A), N, lda, N * lda, B, D.ptr(info), D.ptr(ws), ws.numel(), D.stream_ptr(dev)))
    torch.cuda.synchronize()
    L = torch.tril(A[:, :, :N])
    out[f"{N}x{B}"] = dict(info=info.cpu().tolist(), sum=float(L.sum().item()), sq=float((L * L).sum().item()),
                           diag=L.diagonal(dim1=1, dim2=2).sum(dim=1).cpu().tolist())
print(json.dumps(out))
"""


def test_chain_tasks_claimed_only_by_the_rescue_of_the_waits_give_the_same_factors():
    """SF_DF_MISS_CLAIMS: while the queues hold tasks the dispenser claims no chain / front task at all, so every one of them
    is claimed from inside a queued task's wait (after 500 us), run there, and the queued task is resumed -- the path that
    closes the missed-claim window of the dispenser.  Same arithmetic in the same order: identical factors."""
    want, _ = _run(_POTRF)
    got, _ = _run(_POTRF, SF_DF_MISS_CLAIMS="1")
    for key in want:
        assert got[key]["info"] == [0] * len(got[key]["info"])
        assert got[key] == want[key], key


_STALL = r"""
import json, time, warnings, numpy as np, torch
from starfish_amd import _device as D, _lib, synth
lib = _lib.require_gpu()
dev = D.device_of()
N, B = 2048, 4
lda = N + 16
g = torch.Generator(device=dev).manual_seed(7)
A = torch.empty((B, N, lda), dtype=torch.float64, device=dev).normal_(generator=g) * 0.01
A[:, :, :N] = A[:, :, :N] + A[:, :, :N].transpose(1, 2) + torch.eye(N, dtype=torch.float64, device=dev) * 4.0
info = torch.empty((B,), dtype=torch.int32, device=dev)
ws = D.workspace(lib.sf_potrf_workspace_bytes(N, B), dev)
assert lib.sf_debug_cholesky_sequence(4) == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
_lib.check(lib.sf_potrf_batch(D.ptr(A), N, lda, N * lda, B, D.ptr(info), D.ptr(ws), ws.numel(), D.stream_ptr(dev)))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = D.persistent_status(lib)
lib.sf_debug_cholesky_sequence(-1)
# ... and through the product: the first batch of a model meets the same stall, warns, is re-run
o = synth.make_order(N=1024, m=4, seed=5)
model = synth.build_model(o)
P = synth.walker_ball(o, B=16, seed=3)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    t0 = time.perf_counter()
    lnl, minfo = model.log_likelihood_batch(P, return_info=True)
    dt_model = time.perf_counter() - t0
    t0 = time.perf_counter()
    again = model.log_likelihood_batch(P)
    dt_again = time.perf_counter() - t0
print(json.dumps(dict(info=info.cpu().tolist(), seconds=dt, status=st, lnl=lnl.tolist(), again=again.tolist(), minfo=minfo.tolist(),
                      dt_model=dt_model, dt_again=dt_again, enabled=lib.sf_persistent_potrf(-1),
                      warned=[str(x.message) for x in w if issubclass(x.category, RuntimeWarning)])))
"""


def test_a_launch_whose_claimed_task_never_runs_is_given_up_within_the_stall_bound_not_after_four_seconds():
    """What a device shared with other processes does to the persistent kernel (profiles/r05_g_shared_gpu_abort.txt): a workgroup
    that has claimed a task is kept from running, everything downstream waits.  Forced here (tuning build, SF_DF_MISS_CLAIMS=2:
    the chain task of panel 2 of matrix 0 is claimed and never run).  Round 5 gave such a launch up after the 4-s bound of a
    single wait; now the waits watch the launch's progress counter and give up once NO task has completed for 25 ms.  The
    record of the abort (sf_persistent_potrf_status) says why; the host layer's fall-back makes the model's first call cost
    well under a second and warns once."""
    got, _ = _run(_STALL, SF_DF_MISS_CLAIMS="2")
    assert got["info"] == [-5] * 4
    st = got["status"]
    assert st["aborted_launches"] == 1 and st["reason"] == 2, st   # no task completed for 25 ms
    assert st["workgroups_started"] == st["grid"] > 0, st          # (an exclusive box: the whole grid was resident)
    assert st["tasks_completed"] > 0, st                          # (the other three matrices ran to their ends)
    assert got["seconds"] < 0.5, got["seconds"]                    # 25-35 ms of stall + the launch, not 4 s
    assert got["minfo"] == [0] * 16 and np.isfinite(got["lnl"]).all()
    assert len(got["warned"]) == 1 and "no task of the launch completed for 25 ms" in got["warned"][0], got["warned"]
    assert "disabled for this whole process" in got["warned"][0]
    assert got["enabled"] == 0 and got["lnl"] == got["again"]
    assert got["dt_model"] < 1.0, got["dt_model"]
    clean, _ = _run(_MODEL.replace("lib = _lib.require_gpu()", "lib = _lib.require_gpu(); lib.sf_persistent_potrf(0)"))
    assert got["lnl"] == clean["lnl"]  # the values of the launch sequences, bit for bit


def test_n16384_forced_sequence_4_really_launches_the_persistent_kernel():
    """Round 5's fit check counted 129 panels for N = 16384 (64 rows of shifted frame added unconditionally), so the matrix of
    cfg 5 never took the persistent kernel although its tables hold 127 stages (advisor, round 5).  The library counts its
    persistent launches: one more after this call, none when the sequence is switched off; factors equal to the fused
    sequence's to rounding."""
    import torch
    from starfish_amd import _device as D, _lib

    lib = _lib.require_gpu()
    dev = D.device_of()
    N, B = 16384, 2
    lda = N + 16
    g = torch.Generator(device=dev).manual_seed(11)
    # (only the lower triangle is referenced: random entries of standard deviation 0.003 below a diagonal of 4 -- the
    # symmetric matrix they define has spectral radius ~ 4 +- 0.8)
    A0 = torch.empty((B, N, lda), dtype=torch.float64, device=dev).normal_(generator=g) * 0.003
    for b in range(B):
        A0[b, :, :N].diagonal().add_(4.0 + 0.1 * b)
    info = torch.empty((B,), dtype=torch.int32, device=dev)
    ws = D.workspace(lib.sf_potrf_workspace_bytes(N, B), dev)
    out = {}
    try:
        for seq in (4, 0):
            A = A0.clone()
            assert lib.sf_debug_cholesky_sequence(seq) in (-1, 0, 4)
            before = D.persistent_status(lib)["launches"]
            _lib.check(lib.sf_potrf_batch(D.ptr(A), N, lda, N * lda, B, D.ptr(info), D.ptr(ws), ws.numel(), D.stream_ptr(dev)))
            torch.cuda.synchronize()
            assert info.cpu().tolist() == [0] * B
            out[seq] = (D.persistent_status(lib)["launches"] - before, torch.tril(A[:, :, :N]).clone())
            del A
    finally:
        lib.sf_debug_cholesky_sequence(-1)
    assert out[4][0] == 1 and out[0][0] == 0
    L4, L0 = out[4][1], out[0][1]
    assert float((L4 - L0).abs().max()) <= 1e-11 * float(L0.abs().max())
    # L L^T = A on a row block (the factor itself, not only agreement between two sequences)
    r0, r1 = 9000, 9128
    want = torch.tril(A0[0, :, :N])[r0:r1, :].clone()
    want[:, r0:] = torch.tril(A0[0, r0:, r0:N]).T[: r1 - r0, :]  # (columns right of the diagonal: the transposed lower part)
    want[:, r0:r1] = torch.tril(A0[0, r0:r1, r0:r1]) + torch.tril(A0[0, r0:r1, r0:r1], -1).T
    got = L4[0, r0:r1, :] @ L4[0].T
    assert float((got - want).abs().max()) < 1e-11
