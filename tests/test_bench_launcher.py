"""`python bench.py --gpus N`: the script starts its own N ranks (one process per GPU), fails loudly when the GPUs
are not there, and prints ONE line with n_gpus == N.  The 2-rank run on the test box's single GPU uses the
SF_BENCH_RANKS_SHARE_GPU=1 hook (every rank on device 0, gloo for the barrier -- RCCL refuses two ranks per device)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SF_BENCH_RANKS_SHARE_GPU"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, env=e, timeout=timeout)


def _gpus():
    import torch

    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def test_gpus_2_without_two_gpus_fails_loudly():
    """No silent one-rank run: with fewer than N GPUs visible `--gpus N` exits non-zero and says why."""
    if _gpus() >= 2:
        pytest.skip("two GPUs are visible")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "--gpus 2 needs 2 visible GPUs" in r.stderr
    assert not r.stdout.strip()


def test_gpus_flag_must_match_the_launched_world_size():
    """Under somebody else's launcher (torch.distributed.run) the flag is checked against WORLD_SIZE."""
    env = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1"}
    r = _run(["--gpus", "4"], env=env)
    assert r.returncode == 2 and "launcher started 2 rank(s)" in r.stderr
    r = _run(["--gpus", "1"], env=env)
    assert r.returncode == 2 and "launcher started 2 rank(s)" in r.stderr


def test_order_major_slices_cover_the_unit_list():
    from starfish_amd.parallel import order_major_slices, shard_range

    for n_orders, B, world in ((25, 64, 8), (25, 3, 2), (3, 5, 2), (2, 1, 4)):
        seen = []
        for r in range(world):
            lo, hi = shard_range(n_orders * B, r, world)
            sl = order_major_slices(n_orders, B, lo, hi)
            assert sum(b - a for _, a, b in sl) == hi - lo
            assert all(0 <= a < b <= B for _, a, b in sl)
            seen += [o * B + k for o, a, b in sl for k in range(a, b)]
        assert seen == list(range(n_orders * B))


@pytest.mark.gpu
def test_bench_gpus_2_spawns_two_ranks_and_reports_both_scalings():
    if _gpus() < 1:
        pytest.skip("no GPU")
    r = _run(["--gpus", "2", "--npix", "256", "--batch", "32", "--steps", "2", "--warmup", "1", "--cpu-sample", "0",
              "--no-structured"], env={"SF_BENCH_RANKS_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_batch"] == 64 and d["config"]["units_per_gpu"] == 32
    s = d["strong"]
    assert s["scaling"] == "strong" and s["n_gpus"] == 2 and s["global_batch"] == 32 and s["units_per_gpu"] == 16
    assert s["value"] > 0 and s["ms_per_step"] > 0
    # the clock probe survives the process group (sampled before it is created); its value means nothing for a step
    # this short (the probe wave mostly sees an idle chip)
    assert d["roofline"]["sustained_clock_mhz"] and d["roofline"]["sustained_clock_mhz"] > 0
    assert "before the process group" in d["roofline"]["sustained_clock_note"]


@pytest.mark.gpu
def test_bench_gpus_2_on_a_one_gpu_box_fails_loudly():
    if _gpus() != 1:
        pytest.skip("needs exactly one visible GPU")
    r = _run(["--gpus", "2", "--npix", "256", "--batch", "32", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "needs 2 visible GPUs, found 1" in r.stderr and not r.stdout.strip()
