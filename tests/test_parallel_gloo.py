"""World-size-2 gloo tests of the multi-GPU sharding helpers (CPU): the N > 1 path has no data-path
collective, only a host gather of B doubles."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from starfish_amd.parallel import gather_host, order_major_units, shard_range, sharded_batch


def test_shard_range_partitions_exactly():
    for n in (0, 1, 5, 128, 1600, 1601):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                seen.extend(range(lo, hi))
            assert seen == list(range(n))
            sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_order_major_units():
    u = order_major_units(3, 4)
    assert u.shape == (12, 2) and u[0].tolist() == [0, 0] and u[4].tolist() == [1, 0] and u[-1].tolist() == [2, 3]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        P = rng.standard_normal((B, 13))
        calls = []

        def evaluate(Ps):  # stands in for model.log_likelihood_batch on this rank's GPU
            calls.append(len(Ps))
            return -0.5 * (Ps**2).sum(axis=1)

        full = sharded_batch(evaluate, P)
        lo, hi = shard_range(B, rank, world)
        assert calls == ([hi - lo] if hi > lo else [])
        np.testing.assert_allclose(full, -0.5 * (P**2).sum(axis=1), rtol=0, atol=0)
        # multi-order sum on the host: units are (order, walker) pairs, order-major
        units = order_major_units(3, B)
        ulo, uhi = shard_range(len(units), rank, world)
        local = np.array([float(o * 1000 + w) for o, w in units[ulo:uhi]])
        allu = gather_host(local, len(units))
        per_walker = allu.reshape(3, B).sum(axis=0)
        np.testing.assert_allclose(per_walker, 3000.0 + 3 * np.arange(B))
        np.save(os.path.join(out_dir, f"ok{rank}.npy"), full)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [128, 5, 1])
def test_sharded_batch_world_size_2_gloo(tmp_path, B):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), B, str(tmp_path)), nprocs=world, join=True)
    a = np.load(tmp_path / "ok0.npy")
    b = np.load(tmp_path / "ok1.npy")
    np.testing.assert_array_equal(a, b)
    assert a.shape == (B,)
