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


def _line(r):
    """The one JSON line of a bench run; a failed run's `error` field (and the end of stderr) in the assertion message."""
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout[-800:], r.stderr[-800:])
    d = json.loads(lines[0])
    assert r.returncode == 0 and "error" not in d, (d.get("error"), r.stderr[-1200:])
    return d


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
    # (two processes time-slice the one GPU here: a probe wave that was context-switched is discarded by bench.py)
    clk = d["roofline"]["sustained_clock_mhz"]
    assert clk is None or 200 < clk < 4000
    if clk is not None:
        assert "before the process group" in d["roofline"]["sustained_clock_note"]


@pytest.mark.gpu
def test_bench_gpus_2_on_a_one_gpu_box_fails_loudly():
    if _gpus() != 1:
        pytest.skip("needs exactly one visible GPU")
    r = _run(["--gpus", "2", "--npix", "256", "--batch", "32", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "needs 2 visible GPUs, found 1" in r.stderr and not r.stdout.strip()


def test_eight_way_order_major_split_of_cfg4():
    """cfg 4 = the 1600 (order x walker) units of cfg 3 over 8 ranks: 200 units each, every rank owns walkers of at most
    four orders (so at most four orders' static data per GPU), whole orders except at the ends of its slice, and the
    ranks' slices tile the unit list in order (docs/intro.rst:71-73: orders / chains are independent)."""
    from starfish_amd.parallel import order_major_slices, shard_range

    n_orders, B, world = 25, 64, 8
    owned = []
    for r in range(world):
        lo, hi = shard_range(n_orders * B, r, world)
        assert hi - lo == 200
        sl = order_major_slices(n_orders, B, lo, hi)
        orders = [o for o, _, _ in sl]
        assert orders == sorted(set(orders)) and len(orders) <= 4
        assert orders == list(range(orders[0], orders[-1] + 1))            # a contiguous run of orders
        assert all((a, b) == (0, B) for _, a, b in sl[1:-1])              # whole orders in the middle
        assert sl[0][0] * B + sl[0][1] == lo and sl[-1][0] * B + sl[-1][2] == hi
        owned.append(orders)
    assert owned[0][0] == 0 and owned[-1][-1] == n_orders - 1
    # an order is shared by at most two neighbouring ranks
    for o in range(n_orders):
        holders = [r for r in range(world) if o in owned[r]]
        assert 1 <= len(holders) <= 2 and holders == list(range(holders[0], holders[-1] + 1))


def test_rank_0_prints_one_parsable_line_when_the_run_fails():
    """Without a GPU (this container) the run cannot start: the exit code says so AND stdout carries exactly one JSON
    line with an `error` key and the contract's fields, so a driver that only parses stdout sees why."""
    if _gpus() > 0:
        pytest.skip("a GPU is visible: the failure path is covered by the -m gpu launcher tests")
    r = _run(["--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 1 and d["unit"] == "evals/s" and "error" in d
    assert "MI355X" in d["error"]


_SMALL = ["--npix", "512", "--batch", "16", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--no-structured"]
_ONE_RANK = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "SF_BENCH_FORCE_GROUP": "1"}


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


@pytest.mark.gpu
def test_bench_runs_the_real_rccl_branch_with_one_rank():
    """The N > 1 code path on the one-GPU box: RCCL accepts a single rank, so init (gloo control group + nccl group on
    the device bound by LOCAL_RANK), the trial all-reduce, barrier, max-over-ranks, the clock-probe-before-the-group
    order and the second scaling leg all execute for real."""
    if _gpus() < 1:
        pytest.skip("no GPU")
    r = _run(["--gpus", "1"] + _SMALL, env=dict(_ONE_RANK, MASTER_PORT=_free_port()))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["process_group"].startswith("nccl"), d["process_group"]
    assert d["n_gpus"] == 1 and d["value"] > 0 and "error" not in d
    assert d["strong"]["value"] > 0 and d["strong"]["units_per_gpu"] == 16
    assert "before the process group" in d["roofline"]["sustained_clock_note"]


@pytest.mark.gpu
def test_bench_falls_back_to_gloo_when_rccl_cannot_be_brought_up():
    if _gpus() < 1:
        pytest.skip("no GPU")
    r = _run(["--gpus", "1"] + _SMALL, env=dict(_ONE_RANK, MASTER_PORT=_free_port(), SF_BENCH_FAIL_NCCL="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["process_group"].startswith("gloo (fallback") and d["value"] > 0
    assert "RCCL group unavailable" in r.stderr


@pytest.mark.gpu
def test_a_failing_rank_still_leaves_one_parsable_line_from_rank_0():
    """Rank 1 dies before the process group exists; the launcher terminates rank 0, which may sit in the rendezvous:
    its watchdog thread prints the error line (with rank 1's message) and the launcher's exit code is non-zero."""
    if _gpus() < 1:
        pytest.skip("no GPU")
    r = _run(["--gpus", "2"] + _SMALL, env={"SF_BENCH_RANKS_SHARE_GPU": "1", "SF_BENCH_FAIL_RANK": "1"}, timeout=900)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and "error" in d
    assert "rank1" in d["error"] and "SF_BENCH_FAIL_RANK" in d["error"]


def test_watchdog_thread_prints_the_error_line_on_sigterm(tmp_path):
    """Rank 0 blocked (here: in a lock; on the node: inside a collective or the rendezvous) when the launcher
    terminates it: the wake-up byte of SIGTERM reaches the watchdog thread, which prints the line with the failed
    rank's message and exits 143.  Pure host logic: runs without a GPU."""
    import signal
    import time

    port = _free_port()
    script = tmp_path / "blocked_rank0.py"
    script.write_text(f"""
import argparse, os, sys, tempfile, threading
sys.path.insert(0, {ROOT!r})
import bench
os.environ["MASTER_PORT"] = {port!r}
args = argparse.Namespace(config="cfg2", npix=None, batch=None, steps=1, warmup=0, scaling="weak")
line = bench.ErrorLine(args, 0, 2)
with open(os.path.join(line.dir, "rank1.err"), "w") as fh:
    fh.write("rank 1: RuntimeError: boom")
sys.stderr.write("ready\\n"); sys.stderr.flush()
lock = threading.Lock(); lock.acquire(); lock.acquire()
""")
    p = subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.stderr.readline().strip() == "ready"
    time.sleep(0.2)
    p.send_signal(signal.SIGTERM)
    out, _ = p.communicate(timeout=60)
    assert p.returncode == 143
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] is None and "rank1: rank 1: RuntimeError: boom" in d["error"]


@pytest.mark.gpu
def test_eight_rank_rehearsal_on_one_gpu_cfg2_strong_and_cfg4():
    """`bench.py --gpus 8` has never run on eight GPUs here (one GPU per box): the eight PROCESSES can still be rehearsed --
    eight contexts on the one device (SF_BENCH_RANKS_SHARE_GPU=1, gloo), eight slices, one line with n_gpus == 8.
    (a) cfg 2 strong: 128 walkers -> 16 per rank (the persistent-kernel batch of an 8-way split), weak sub-object 8 x 128;
    (b) cfg 4 = cfg 3 strong, order-major: 1600 units -> 200 per rank, each rank builds at most four orders."""
    if _gpus() < 1:
        pytest.skip("no GPU")
    hook = {"SF_BENCH_RANKS_SHARE_GPU": "1", "OMP_NUM_THREADS": "4"}
    r = _run(["--gpus", "8", "--scaling", "strong", "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--no-structured"],
             env=hook, timeout=1500)
    d = _line(r)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["global_batch"] == 128 and d["config"]["units_per_gpu"] == 16
    wk = d["weak"]
    assert wk["n_gpus"] == 8 and wk["global_batch"] == 8 * 128 and wk["units_per_gpu"] == 128 and wk["value"] > 0
    # the 128 walkers of the strong split are the walkers of the one-rank batch: same sum of lnL (16 matrices per call
    # instead of 128: another launch sequence, equal to rounding)
    one = _run(["--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--no-structured", "--no-extra-legs"], timeout=900)
    d1 = _line(one)
    assert d1["n_gpus"] == 1 and abs(d["lnl_checksum"] - d1["lnl_checksum"]) <= 1e-11 * abs(d1["lnl_checksum"])

    r = _run(["--gpus", "8", "--config", "cfg3", "--scaling", "strong", "--single-scaling", "--steps", "1", "--warmup", "1",
              "--cpu-sample", "0"], env=hook, timeout=1500)
    d = _line(r)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["global_batch"] == 1600 and d["config"]["units_per_gpu"] == 200
    assert d["lnl_checksum"] == d["lnl_checksum"] and abs(d["lnl_checksum"]) > 0  # (finite: every unit of every rank evaluated)
