"""Per-queue busy time and per-kernel-family statistics of one sf_potrf_batch call traced by tools/trace_potrf.sh:
python tools/trace_summary.py [gpurun_out/trace_potrf]"""
import collections
import csv
import glob
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace_potrf"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in ("k_diag", "k_chol_panel", "k_gemm_nt", "k_potrf_dataflow"))]
bursts, cur = [], [idx[0]]
for a, b in zip(idx, idx[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]) > 2_000_000:
        bursts.append(cur)
        cur = []
    cur.append(b)
bursts.append(cur)
sel = bursts[1] if len(bursts) > 1 else bursts[0]
t0 = int(rows[sel[0]]["Start_Timestamp"])
t1 = max(int(rows[i]["End_Timestamp"]) for i in sel)
print(f"{len(sel)} launches, {(t1 - t0) / 1e6:.3f} ms")
fam = collections.defaultdict(lambda: [0, 0.0, 1e18, 0.0])
busy = collections.defaultdict(float)
for i in sel:
    r = rows[i]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wg = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    name = r["Kernel_Name"].replace("void ", "").split("(")[0]
    key = (r["Queue_Id"], name)
    e = fam[key]
    e[0] += 1
    e[1] += dur
    e[2] = min(e[2], dur)
    e[3] = max(e[3], dur)
    busy[r["Queue_Id"]] += dur
for q in sorted(busy):
    print(f"queue {q}: busy {busy[q] / 1e3:.3f} ms ({busy[q] * 1e3 / (t1 - t0) * 100:.0f} %)")
for (q, name), e in sorted(fam.items()):
    print(f"  q{q} {name:40s} n={e[0]:4d} sum={e[1] / 1e3:7.3f} ms  avg={e[1] / e[0]:7.1f} us  min={e[2]:7.1f} max={e[3]:7.1f}")
if len(sys.argv) > 2:
    lo, hi = float(sys.argv[2]) * 1e3, float(sys.argv[3]) * 1e3
    for i in sel:
        r = rows[i]
        st, en = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        if lo <= st <= hi:
            wg = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
            print(f"{st / 1e3:9.1f} {(en - st) / 1e3:7.1f} {en / 1e3:9.1f} q{r['Queue_Id']} wg={wg:5d} {r['Kernel_Name'].replace('void ', '')[:30]}")
