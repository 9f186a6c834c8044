"""Summarise rocprofv3 output (kernel stats + separate --pmc passes) of `bench.py` into the small files
committed under profiles/:  python tools/summarize_profile.py gpurun_out/r1_final profiles/r01_b
PMC passes ran `bench.py --steps 1 --warmup 1`: counters are summed over the dispatches of the LAST
step only.  FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for
gfx950 wide coalesced reads; WRITE_SIZE is taken as reported (KB)."""
import collections
import csv
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]


def short(name):
    return name.split("(")[0].replace("void ", "")[:40]


out = {}
stats = list(csv.DictReader(open(os.path.join(src, "stats", "bench_kernel_stats.csv"))))
with open(dst + "_kernel_stats.csv", "w") as fh:
    fh.write("kernel,calls,total_ms,avg_us,percent\n")
    for r in stats:
        fh.write(f"{short(r['Name'])},{r['Calls']},{int(r['TotalDurationNs'])/1e6:.3f},"
                 f"{float(r['AverageNs'])/1e3:.2f},{float(r['Percentage']):.2f}\n")

# union of the (overlapping, two-stream) k_gemm_nt launch intervals per bench step, from the kernel trace
trace_f = os.path.join(src, "stats", "bench_kernel_trace.csv")
gemm_union = None
if os.path.exists(trace_f):
    tr = sorted(csv.DictReader(open(trace_f)), key=lambda r: int(r["Start_Timestamp"]))
    nsteps = sum(1 for r in tr if r["Kernel_Name"].startswith("k_logdet_z"))
    iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in tr if "k_gemm_nt" in r["Kernel_Name"]]
    uni, lo, hi = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a <= hi:
            hi = max(hi, b)
        else:
            uni += hi - lo
            lo, hi = a, b
    uni += hi - lo
    gemm_union = {"steps": nsteps, "launches": len(iv), "union_ms_per_step": uni / 1e6 / max(1, nsteps),
                  "sum_ms_per_step": sum(b - a for a, b in iv) / 1e6 / max(1, nsteps)}

agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
dur = collections.defaultdict(float)
sqdur = collections.defaultdict(float)
for d in sorted(os.listdir(src)):
    f = os.path.join(src, d, "pmc_counter_collection.csv")
    if not os.path.exists(f):
        continue
    rows = list(csv.DictReader(open(f)))
    idx = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_emulator"))
    seen = set()
    for r in rows[idx:]:
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (d, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            if d == "FETCH_SIZE":
                calls[k] += 1
                dur[k] += t
            if d.startswith("SQ_VALU_MFMA"):
                sqdur[k] += t
summary = {}
for k, v in agg.items():
    e = {"launches_per_step": calls[k], "ms_per_step_under_pmc": round(dur[k], 3)}
    if "FETCH_SIZE" in v:
        e["fetch_GB_per_step_corrected_x2"] = round(2 * v["FETCH_SIZE"] * 1024 / 1e9, 3)
    if "WRITE_SIZE" in v:
        e["write_GB_per_step"] = round(v["WRITE_SIZE"] * 1024 / 1e9, 3)
    if "TCC_HIT_sum" in v:
        e["l2_hit_rate"] = round(v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 3)
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES") and sqdur[k] > 0:
        # busy cycles summed over the 1024 SIMDs / (SIMDs x kernel time x 2.38 GHz sustained clock)
        e["mfma_pipe_busy_frac"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * sqdur[k] * 1e-3 * 2.38e9), 3)
    summary[k] = e
g = [k for k in summary if k.startswith("k_gemm_nt")]
tot_fetch = sum(summary[k].get("fetch_GB_per_step_corrected_x2", 0) for k in g)
tot_write = sum(summary[k].get("write_GB_per_step", 0) for k in g)
tot_launch = sum(summary[k]["launches_per_step"] for k in g)
summary["_k_gemm_nt_all"] = {
    "launches_per_step": tot_launch,
    "hbm_GB_per_step": round(tot_fetch + tot_write, 3),
    "hbm_bytes_per_launch": (tot_fetch + tot_write) * 1e9 / max(1, tot_launch),
}
if gemm_union:
    summary["_k_gemm_nt_all"]["kernel_trace"] = gemm_union
json.dump(summary, open(dst + "_pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(summary["_k_gemm_nt_all"]))
for k in sorted(summary):
    print(k, summary[k])
