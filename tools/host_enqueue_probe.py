"""Host-side enqueue time of sf_loglike_batch per step, with and without an initialised RCCL process group
(tuning aid).  usage: python tools/host_enqueue_probe.py [dist]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
use_dist = len(sys.argv) > 1 and sys.argv[1] == "dist"
prof = len(sys.argv) > 2 and sys.argv[2] == "prof"
torch.cuda.set_device(0)
if use_dist:
    import torch.distributed as dist
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0)); dist.barrier()
from gpu_helpers import device_order, oracle_order, pack_rows
from starfish_amd import _device as D, synth
o = synth.make_order(N=4096); oo = oracle_order(o); do = device_order(oo)
P = synth.walker_ball(o, B=128); md, rows = pack_rows(do, [synth.vector_to_oracle_params(p) for p in P])
Pd = D.to_dev(rows, do.dev); lnl = D.empty((128,), do.dev); info = D.empty((128,), do.dev, torch.int32)
for _ in range(2): do.loglike_device(md, Pd, lnl, info)
torch.cuda.synchronize()
do.lib.sf_profile_enable(1 if prof else 0)
if use_dist and len(sys.argv) > 3:
    dist.barrier(); torch.cuda.synchronize()
t0 = time.perf_counter(); host = []
for _ in range(5):
    t = time.perf_counter(); do.loglike_device(md, Pd, lnl, info); host.append((time.perf_counter() - t) * 1e3)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) * 1e3
print("dist" if use_dist else "plain", "prof" if prof else "noprof", "host enqueue ms per step:", [round(h, 2) for h in host], "total per step %.2f ms" % (tot / 5))
