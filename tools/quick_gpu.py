import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from starfish_amd import synth, _device as D
from gpu_helpers import device_order, oracle_order, pack_rows
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
o = synth.make_order(N=N)
t=time.time(); oo = oracle_order(o); print("oracle order", time.time()-t)
t=time.time(); do = device_order(oo); print("device order", time.time()-t)
P = synth.walker_ball(o, B=B)
md, rows = pack_rows(do, [synth.vector_to_oracle_params(p) for p in P])
Pd = D.to_dev(rows, do.dev)
lnl = D.empty((B,), do.dev); info = D.empty((B,), do.dev, torch.int32)
do.lib.sf_profile_enable(0)
for it in range(3):
    torch.cuda.synchronize(); t=time.time()
    do.loglike_device(md, Pd, lnl, info)
    torch.cuda.synchronize(); dt=time.time()-t
    print(f"N={N} B={B} iter {it}: {dt*1e3:.1f} ms  -> {B/dt:.1f} evals/s  TF={B*(N**3/3)/dt/1e12:.2f}")
print(lnl[:4].cpu().numpy(), info[:4].cpu().numpy())
import ctypes as C
do.lib.sf_profile_enable(1)
do.loglike_device(md, Pd, lnl, info); torch.cuda.synchronize()
ms = (C.c_double*6)(); fl=C.c_double(); nl=C.c_long(); nc=C.c_long()
do.lib.sf_profile_read(ms, C.byref(fl), C.byref(nl), C.byref(nc))
print("stage ms [transform, fill, gemm, potrf, solve]:", list(ms), "gemm TF:", fl.value/ (ms[2]*1e-3)/1e12, nl.value)
