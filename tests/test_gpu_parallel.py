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


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    from starfish_amd.parallel import gather_host, shard_range, sharded_batch

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # host gather only: any backend does
    try:
        torch.cuda.set_device(0)
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


def test_two_ranks_shard_real_models(tmp_path):
    world = 2
    ctx = mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=False)
    # bounded wait: a box on which two processes cannot bring the one GPU up together must not wedge the suite
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
        pytest.skip("the two GPU worker processes did not finish within 300 s on this box")
    # the same evaluations in this process, unsharded
    o = synth.make_order(N=512, m=4, seed=21)
    want = synth.build_model(o).log_likelihood_batch(synth.walker_ball(o, B=11, seed=4))
    for r in range(world):
        np.testing.assert_allclose(np.load(tmp_path / f"single{r}.npy"), want, rtol=1e-12)
    orders = [synth.make_order(N=256, m=4, seed=30 + k, wave0=5000.0 * 1.02**k) for k in range(3)]
    want = synth.build_echelle(orders).log_likelihood_batch(synth.shared_ball(orders[0], B=5, seed=2))
    for r in range(world):
        np.testing.assert_allclose(np.load(tmp_path / f"multi{r}.npy"), want, rtol=1e-12)
