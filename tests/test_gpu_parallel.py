"""The N > 1 path with REAL models: two ranks (two processes sharing the one GPU of the test box)
each build their SpectrumModel / EchelleModel, evaluate their slice of the (order x walker) units with
``sharded_batch`` and gather on the host -- no data-path collective.  Run with -m gpu."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from starfish_amd import synth

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ranks_share_one_gpu():
    """The ranks of these tests are processes on ONE device (the test box has one GPU).  The persistent-kernel Cholesky spins
    on counters written by its own other workgroups and needs them all resident -- the GPU to itself, one process per GPU as
    in production; processes oversubscribing a device can starve each other's launches until the 4-s bound aborts them
    (measured with eight bench ranks, profiles/r05_g_shared_gpu_abort.txt; the product then warns and falls back, see
    tests/test_gpu_recovery.py).  Not what these tests are about: switched off up front."""
    from starfish_amd import _lib

    _lib.require_gpu().sf_persistent_potrf(0)


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    from starfish_amd.parallel import gather_host, shard_range, sharded_batch

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # host gather only: any backend does
    try:
        torch.cuda.set_device(0)
        _ranks_share_one_gpu()
        # single order: every rank evaluates its contiguous slice of the walkers
        o = synth.make_order(N=512, m=4, seed=21)
        model = synth.build_model(o)
        P = synth.walker_ball(o, B=11, seed=4)
        full = sharded_batch(model.log_likelihood_batch, P)
        np.save(os.path.join(out_dir, f"single{rank}.npy"), full)
        # multi-order, order-major units (cfg 4): rank r owns a contiguous slice of the (order, walker) list
        orders = [synth.make_order(N=256, m=4, seed=30 + k, wave0=5000.0 * 1.02**k) for k in range(3)]
        em = synth.build_echelle(orders)
        Ps = synth.shared_ball(orders[0], B=5, seed=2)
        n_units = len(orders) * len(Ps)
        lo, hi = shard_range(n_units, rank, world)
        local = []
        for k, m in enumerate(em.orders):
            wlo, whi = max(lo, k * len(Ps)) - k * len(Ps), min(hi, (k + 1) * len(Ps)) - k * len(Ps)
            if whi > wlo:
                local.append(m.log_likelihood_batch(Ps[wlo:whi]))
        local = np.concatenate(local) if local else np.zeros(0)
        units = gather_host(local, n_units)
        np.save(os.path.join(out_dir, f"multi{rank}.npy"), units.reshape(len(orders), len(Ps)).sum(axis=0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_shard_real_models(tmp_path, world):
    """world = 8: the rank count of the BASELINE metric's last column -- eight processes, eight contexts on the one GPU of
    the test box (11 walkers -> one or two per rank: every rank's call is a persistent-kernel launch of 1-2 matrices)."""
    ctx = mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=False)
    # bounded wait: a wedged N > 1 path fails the test (it must not wedge the suite, and it must not pass as a skip)
    import time

    deadline = time.time() + 300
    done = False
    while time.time() < deadline:
        done = ctx.join(timeout=5)  # raises if a worker failed
        if done:
            break
    if not done:
        for p in ctx.processes:
            p.terminate()
        pytest.fail(f"the {world} GPU worker processes did not finish within 300 s: the N > 1 path hangs")
    # the same evaluations in this process, unsharded
    o = synth.make_order(N=512, m=4, seed=21)
    want = synth.build_model(o).log_likelihood_batch(synth.walker_ball(o, B=11, seed=4))
    for r in range(world):
        np.testing.assert_allclose(np.load(tmp_path / f"single{r}.npy"), want, rtol=1e-12)
    orders = [synth.make_order(N=256, m=4, seed=30 + k, wave0=5000.0 * 1.02**k) for k in range(3)]
    want = synth.build_echelle(orders).log_likelihood_batch(synth.shared_ball(orders[0], B=5, seed=2))
    for r in range(world):
        np.testing.assert_allclose(np.load(tmp_path / f"multi{r}.npy"), want, rtol=1e-12)


# ------------------------------------------------------------------------------------------------ cfg 4
def _cfg4_worker(rank, world, port, out_dir):
    """BASELINE cfg 4 shape: the (order x walker) units of the 25 x 3000 model split ORDER-MAJOR over the ranks
    (docs/intro.rst:71-73: orders are independent), each rank builds only the orders it owns and evaluates its
    slice in one MultiPlan pass (sf_loglike_multi_batch); host gather, no collective."""
    import torch
    import torch.distributed as dist

    from conftest import load_golden
    from starfish_amd import _device as D
    from starfish_amd.parallel import gather_host, order_major_slices, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        _ranks_share_one_gpu()
        g = load_golden("model_cfg3.npz")
        n_orders, N, P = int(g["n_orders"][0]), int(g["N"][0]), g["P"]
        B = len(P)
        orders = synth.make_echelle(n_orders, N, seed0=int(g["seed0"][0]))
        lo, hi = shard_range(n_orders * B, rank, world)
        devs, rows_list, md = [], [], None
        slices = order_major_slices(n_orders, B, lo, hi)
        for o, wlo, whi in slices:
            m = synth.build_model(orders[o], freeze=("local_cov",))  # only the orders this rank owns
            d_o, md, rows = m._pack(P[wlo:whi], update_caches=False)
            devs.append(d_o)
            rows_list.append(rows)
        plan = D.MultiPlan(devs, md, rows_list)
        assert plan.units == hi - lo
        plan.enqueue()
        outs = plan.collect()
        assert all((o["info"] == 0).all() for o in outs)
        local = np.concatenate([o["lnl"] for o in outs])
        units = gather_host(local, n_orders * B)
        np.save(os.path.join(out_dir, f"cfg4_{rank}.npy"), units.reshape(n_orders, B))
        np.save(os.path.join(out_dir, f"cfg4_owned_{rank}.npy"), np.array([o for o, _, _ in slices]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_cfg4_order_major_split_vs_reference_goldens(tmp_path, world):
    """cfg 4 = cfg 3 sharded: 25 orders x N = 3000, the three golden walkers, two ranks -- and EIGHT, the rank count cfg 4
    names (75 units -> 9 or 10 per rank, every rank builds only the 4-5 orders it owns); the gathered per-order lnL
    must equal the REFERENCE's per-order values (tests/golden/model_cfg3.npz) and their sum the model lnL."""
    import time

    from conftest import load_golden
    from starfish_amd.parallel import order_major_slices, shard_range

    ctx = mp.spawn(_cfg4_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=False)
    deadline = time.time() + 600
    done = False
    while time.time() < deadline:
        done = ctx.join(timeout=5)
        if done:
            break
    if not done:
        for p in ctx.processes:
            p.terminate()
        pytest.fail(f"the {world} GPU worker processes did not finish within 600 s: the N > 1 path hangs")
    g = load_golden("model_cfg3.npz")
    want = g["lnl"]
    for r in range(world):
        got = np.load(tmp_path / f"cfg4_{r}.npy")
        assert got.shape == want.shape
        assert np.all(np.abs(got - want) <= 1e-8 * np.abs(want) + 1e-8), np.max(np.abs(got - want) / np.abs(want))
        np.testing.assert_allclose(got.sum(axis=0), want.sum(axis=0), rtol=1e-9)
    # order-major: with two ranks rank 0 owns orders 0..12 (12 shared with rank 1), rank 1 owns 12..24
    owned = [np.load(tmp_path / f"cfg4_owned_{r}.npy").tolist() for r in range(world)]
    if world == 2:
        assert owned[0] == list(range(13)) and owned[1] == list(range(12, 25))
    n_orders, B = want.shape
    for r in range(world):
        lo, hi = shard_range(n_orders * B, r, world)
        assert owned[r] == [o for o, _, _ in order_major_slices(n_orders, B, lo, hi)]
        assert owned[r] == list(range(owned[r][0], owned[r][-1] + 1)) and len(owned[r]) <= -(-n_orders // world) + 1
    assert owned[0][0] == 0 and owned[-1][-1] == n_orders - 1


# ------------------------------------------------------------- eight ranks, each with the device to itself in turn
def _turns_worker(rank, world, port, out_dir):
    """cfg 2 split eight ways (16 walkers per rank) with the PER-GPU DEFAULT path: the persistent-kernel Cholesky stays on.
    Eight GPUs give every rank a device of its own; the one GPU of the test box is handed round instead -- an exclusive file
    lock around every device call, so that no two processes ever have kernels in flight together (the eight contexts and
    their memory stay resident side by side: that much of a shared device remains)."""
    import fcntl
    import json
    import warnings

    import torch
    import torch.distributed as dist

    from starfish_amd import _device as D, _lib
    from starfish_amd.parallel import gather_host, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        lib = _lib.require_gpu()
        assert lib.sf_persistent_potrf(-1) == 1  # the library's own choice, not switched off
        o = synth.make_order(N=4096)
        P = synth.walker_ball(o, B=128)
        lo, hi = shard_range(len(P), rank, world)
        lock = open(os.path.join(out_dir, "device.lock"), "w")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            fcntl.flock(lock, fcntl.LOCK_EX)  # ---- this rank's turn: build (init-time device work) + evaluate + synchronise
            try:
                model = synth.build_model(o)
                before = D.persistent_status(lib)
                local, info = model.log_likelihood_batch(P[lo:hi], return_info=True)
                again = model.log_likelihood_batch(P[lo:hi])
                torch.cuda.synchronize()
                after = D.persistent_status(lib)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
        assert (info == 0).all()
        np.testing.assert_array_equal(local, again)
        full = gather_host(local, len(P))  # host gather (gloo), no data-path collective
        np.save(os.path.join(out_dir, f"turns{rank}.npy"), full)
        with open(os.path.join(out_dir, f"turns{rank}.json"), "w") as fh:
            json.dump(dict(launches=after["launches"] - before["launches"], aborted=after["aborted_launches"],
                           enabled=lib.sf_persistent_potrf(-1), warned=[str(x.message) for x in w]), fh)
    finally:
        dist.destroy_process_group()


def test_eight_ranks_take_turns_on_the_device_with_the_persistent_kernel_on(tmp_path):
    """the shared-device rehearsals above switch the persistent kernel off, so the path eight GPUs would run --
    k_potrf_dataflow on 16 matrices per rank -- had never run inside a multi-process job.  Here it does (two persistent
    launches per rank, none aborted, no warning); the gathered batch equals what ONE process computes slice by slice bit for
    bit, the full batch of 128 (another launch sequence: panel pairs) to rounding, and the reference's values."""
    import json
    import time

    from conftest import load_golden
    from starfish_amd.parallel import shard_range

    world = 8
    ctx = mp.spawn(_turns_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=False)
    deadline = time.time() + 600
    done = False
    while time.time() < deadline:
        done = ctx.join(timeout=5)
        if done:
            break
    if not done:
        for p in ctx.processes:
            p.terminate()
        pytest.fail("the 8 GPU worker processes did not finish within 600 s")
    o = synth.make_order(N=4096)
    P = synth.walker_ball(o, B=128)
    model = synth.build_model(o)
    want_slices = np.concatenate([model.log_likelihood_batch(P[slice(*shard_range(128, r, world))]) for r in range(world)])
    want_full = model.log_likelihood_batch(P)
    g2, gf = load_golden("model_cfg2.npz"), load_golden("model_fullbatch.npz")
    for r in range(world):
        got = np.load(tmp_path / f"turns{r}.npy")
        np.testing.assert_array_equal(got, want_slices)           # same launch shape (16 matrices): same bits
        np.testing.assert_allclose(got, want_full, rtol=1e-11)    # the full batch takes the panel-pair sequence
        assert np.all(np.abs(got[:8] - g2["n4096_batch_lnl"]) <= 1e-8 * np.abs(g2["n4096_batch_lnl"]) + 1e-8)
        assert abs(got[127] - gf["cfg2_lnl127"][0]) <= 1e-8 * abs(gf["cfg2_lnl127"][0]) + 1e-8
        rec = json.load(open(tmp_path / f"turns{r}.json"))
        assert rec["launches"] == 2 and rec["aborted"] == 0 and rec["enabled"] == 1 and rec["warned"] == [], rec
