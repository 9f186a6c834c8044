"""Utilisation timeline of one dataflow launch from its per-task trace (tuning build):
    SF_LIB_PATH=.../libstarfish_amd_tuning.so SF_DF_VERBOSE=1 SF_DF_TRACE_FILE=/tmp/df.txt python tools/bench_potrf.py 4096 32 1 4
    python tools/df_trace.py /tmp/df.txt [bin_us]"""
import sys
import numpy as np

path = sys.argv[1]
bin_us = float(sys.argv[2]) if len(sys.argv) > 2 else 250.0
head = open(path).readline().strip()
d = np.loadtxt(path, dtype=np.int64, ndmin=2)
ty, k, i, b, wg, t0, t1, t2 = d.T[:8]
inw = d.T[8:11] / 100.0 if d.shape[1] >= 11 else np.zeros((3, len(d)))  # (in-body waits, us: written by experiment builds only)
T0 = t0.min()
t0, t1, t2 = (t0 - T0) / 100.0, (t1 - T0) / 100.0, (t2 - T0) / 100.0  # us
nwg = int(wg.max()) + 1
span = t2.max()
print(head)
print(f"{len(d)} tasks, {nwg} workgroups, span {span / 1e3:.3f} ms; bodies {np.sum(t2 - t1) / nwg / 1e3:.3f} ms/wg "
      f"(of which in-body waits: columns {inw[0].sum() / nwg / 1e3:.3f}, diagonal tile {inw[1].sum() / nwg / 1e3:.3f}, step 4 {inw[2].sum() / nwg / 1e3:.3f}), "
      f"waits {np.sum(t1 - t0) / nwg / 1e3:.3f} ms/wg, neither {(span * nwg - np.sum(t2 - t0)) / nwg / 1e3:.3f} ms/wg")
names = ["C", "FP", "FR", "R", "RP", "RR"]
nb = int(span / bin_us) + 1
edges = np.arange(nb + 1) * bin_us


def occupancy(a, z):  # time within every bin covered by the intervals [a, z)
    out = np.zeros(nb)
    for lo, hi in zip(a, z):
        j0, j1 = int(lo / bin_us), min(int(hi / bin_us), nb - 1)
        for j in range(j0, j1 + 1):
            out[j] += max(0.0, min(hi, edges[j + 1]) - max(lo, edges[j]))
    return out / (bin_us * nwg)


body, wait = occupancy(t1, t2), occupancy(t0, t1)
bt = {n: occupancy(t1[ty == j], t2[ty == j]) for j, n in enumerate(names)}
print("   t_ms  busy  wait  idle | share of the busy time by type " + " ".join(f"{n:>4}" for n in names) + " | stages in flight")
for j in range(nb):
    live = (t0 < edges[j + 1]) & (t2 > edges[j])
    ks = k[live]
    print(f"{edges[j] / 1e3:7.2f} {body[j]:5.2f} {wait[j]:5.2f} {1 - body[j] - wait[j]:5.2f} | " +
          " ".join(f"{bt[n][j] / max(body[j], 1e-9):4.2f}" for n in names) + (f" | {ks.min()}..{ks.max()}" if len(ks) else " |"))
print("waiting by stage k (ms per workgroup):")
for kk in range(int(k.max()) + 1):
    m = k == kk
    if m.any():
        print(f"  k={kk:2d}: wait {np.sum((t1 - t0)[m]) / nwg / 1e3:6.3f}  body {np.sum((t2 - t1)[m]) / nwg / 1e3:6.3f}  first claim {t0[m].min() / 1e3:6.2f}  last end {t2[m].max() / 1e3:6.2f} ms;"
 + f" in-body {inw[0][m].sum() / nwg / 1e3:.3f}/{inw[1][m].sum() / nwg / 1e3:.3f}/{inw[2][m].sum() / nwg / 1e3:.3f};"
              + " wait by type " + " ".join(f"{n} {np.sum((t1 - t0)[m & (ty == j)]) / nwg / 1e3:.3f}" for j, n in enumerate(names) if (m & (ty == j)).any()))

print("in-body waits by type (ms per workgroup): columns / diagonal tile / step 4")
for j, n in enumerate(names):
    m = ty == j
    if m.any():
        print(f"  {n:>2}: {inw[0][m].sum() / nwg / 1e3:.3f} / {inw[1][m].sum() / nwg / 1e3:.3f} / {inw[2][m].sum() / nwg / 1e3:.3f};  mean body {np.mean((t2 - t1)[m]):7.1f} us over {m.sum()} tasks")
# gaps between a workgroup's tasks = the dispenser
order = np.lexsort((t0, wg))
gap = (t0[order][1:] - t2[order][:-1])[wg[order][1:] == wg[order][:-1]]
print(f"dispenser: gap between a workgroup's tasks: mean {gap.mean():.1f} us, median {np.median(gap):.1f}, p90 {np.percentile(gap, 90):.1f}, sum {gap.sum() / nwg / 1e3:.3f} ms/wg")
