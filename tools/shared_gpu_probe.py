"""What a device SHARED by several processes does to the persistent-kernel Cholesky, and what the fall-back costs:
    python tools/shared_gpu_probe.py [nproc ...]          (default: 1 2 8)
Every process builds the cfg-2 model (N = 4096) on device 0 and evaluates the same 16 walkers `CALLS` times through
SpectrumModel.log_likelihood_batch with the persistent kernel ON (the library's default for 16 matrices); all processes
start their calls together (file barrier).  Printed per process: the wall time of every call, the warnings, the library's
abort record (sf_persistent_potrf_status) and whether the values equal those of an exclusive process bit for bit / to
rounding.  One process per GPU -- the deployment the metric names -- is the nproc = 1 line."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALLS = 6

WORKER = r"""
import json, os, sys, time, warnings
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from starfish_amd import synth, _device as D, _lib
rank, nproc, sync_dir, calls = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
lib = _lib.require_gpu()
torch.cuda.set_device(0)
o = synth.make_order(N=4096)
model = synth.build_model(o)
P = synth.walker_ball(o, B=16, seed=1)
torch.cuda.synchronize()
open(os.path.join(sync_dir, f"ready{rank}"), "w").close()
while len([f for f in os.listdir(sync_dir) if f.startswith("ready")]) < nproc:
    time.sleep(0.002)
out = dict(rank=rank, ms=[], lnl=None, warned=[])
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    for c in range(calls):
        t0 = time.perf_counter()
        lnl, info = model.log_likelihood_batch(P, return_info=True)
        out["ms"].append((time.perf_counter() - t0) * 1e3)
        assert (info == 0).all(), info
        if out["lnl"] is None:
            out["lnl"] = lnl.tolist()
            out["first_equals_later"] = True
        else:
            out["first_equals_later"] = out["first_equals_later"] and (lnl.tolist() == out["lnl"] or bool(np.allclose(lnl, out["lnl"], rtol=1e-11)))
    out["warned"] = [str(x.message)[:220] for x in w if issubclass(x.category, RuntimeWarning)]
out["status"] = D.persistent_status(lib)
print(json.dumps(out))
"""


def run(nproc):
    with tempfile.TemporaryDirectory() as d:
        procs = [subprocess.Popen([sys.executable, "-c", WORKER, ROOT, str(r), str(nproc), d, str(CALLS)], stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True) for r in range(nproc)]
        outs = []
        for p in procs:
            so, se = p.communicate(timeout=900)
            line = [ln for ln in so.splitlines() if ln.startswith("{")]
            outs.append(json.loads(line[-1]) if line else dict(error=se[-400:]))
    return outs


def main():
    counts = [int(a) for a in sys.argv[1:]] or [1, 2, 8]
    ref = None
    for n in counts:
        t0 = time.time()
        outs = run(n)
        print(f"# {n} process(es) on one device, {CALLS} calls of 16 walkers each (N = 4096), {time.time() - t0:.0f} s in all")
        for o in outs:
            if "error" in o:
                print("  ERROR", o["error"])
                continue
            if ref is None:
                ref = o["lnl"]
            st = o["status"]
            same = "bit-identical to the exclusive run" if o["lnl"] == ref else (
                "equal to the exclusive run to 1e-11 (another launch sequence)" if max(abs(a - b) / abs(b) for a, b in zip(o["lnl"], ref)) < 1e-11 else "DIFFERENT")
            print(f"  rank {o['rank']}: calls (ms) " + " ".join(f"{m:.1f}" for m in o["ms"]) + f" | warnings {len(o['warned'])} | aborted launches "
                  f"{st['aborted_launches']} (reason {st['reason']}, {st['workgroups_started']}/{st['grid']} workgroups started, "
                  f"{st['tasks_completed']} tasks done) | persistent launches {st['launches']}, enabled afterwards {st['enabled']} | first call {same}")
            for wmsg in o["warned"][:1]:
                print("     warning:", wmsg)


if __name__ == "__main__":
    main()
