"""Summarise rocprofv3 output of tools/profile_bench.sh (kernel stats + separate --pmc passes of `bench.py`) into the
small files committed under profiles/:  python tools/summarize_profile.py gpurun_out/<tag> profiles/<tag>
The PMC passes ran `bench.py --steps 1 --warmup 1`, i.e. TWO identical steps: counters are summed over all
dispatches of a kernel and divided by two.  FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM
section) prescribes for gfx950 wide coalesced reads; WRITE_SIZE is taken as reported (KB)."""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
PMC_STEPS = 2
PANEL = ("k_chol_panel", "k_gemm_nt", "k_potrf_dataflow")  # the fp64 MFMA kernels of the batched Cholesky


def short(name):
    return name.split("(")[0].replace("void ", "")[:40]


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


def bench_line(name):
    """Last JSON line a bench.py run of the profiling script printed (bench_under_rocprof.json / bench_plain.json)."""
    try:
        with open(os.path.join(src, name)) as fh:
            return json.loads([ln for ln in fh if ln.startswith("{")][-1])
    except Exception:
        return {}


def interval_union(iv):
    iv = sorted(iv)
    if not iv:
        return 0
    uni, lo, hi = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a <= hi:
            hi = max(hi, b)
        else:
            uni += hi - lo
            lo, hi = a, b
    return uni + hi - lo


stats_f = find("stats/**/*kernel_stats.csv")
stats = list(csv.DictReader(open(stats_f))) if stats_f else []

# union of the (overlapping, multi-stream) launch intervals of a kernel FAMILY, from the kernel trace of the same run:
# per-kernel total_ms sums launches that run side by side on three streams (cfg 2: 772 ms summed, 321 ms elapsed), so
# the per-kernel rows alone overstate the time; the `_union:` rows below are what a roofline fraction is computed from
trace_f = find("stats/**/*kernel_trace.csv")
union, union_rows = None, []
if trace_f:
    tr = sorted(csv.DictReader(open(trace_f)), key=lambda r: int(r["Start_Timestamp"]))
    nfinish = max(1, sum(1 for r in tr if r["Kernel_Name"].startswith("k_finish")))
    iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in tr if any(p in r["Kernel_Name"] for p in PANEL)]
    under = bench_line("bench_under_rocprof.json")
    roof = under.get("roofline") or {}
    # steps of the traced run: what its own line says (warm-up included); a multi-order step has one k_finish per CHUNK
    nsteps = int(under["steps"] + under["warmup"]) if "steps" in under and "warmup" in under else nfinish
    if iv:
        uni = interval_union(iv)
        union = {"k_finish_calls": nfinish, "steps": nsteps, "launches": len(iv), "union_ms_total": uni / 1e6,
                 "sum_ms_total": sum(b - a for a, b in iv) / 1e6}
        # algorithmic flops of the panel launches of ONE step, as the library counts them (bench line of the same run)
        gf = None
        if roof.get("algorithmic_flops_per_launch") and roof.get("launches") and roof.get("profiled_steps"):
            gf = roof["algorithmic_flops_per_launch"] * roof["launches"] / roof["profiled_steps"] / 1e9
        ms_step = uni / 1e6 / nsteps
        union_rows.append(("_union:panel kernels (k_chol_panel* + k_potrf_dataflow + k_gemm_nt)", len(iv), uni / 1e6, nsteps,
                           ms_step, gf, "GFLOP", gf / ms_step if gf else None, "TFLOP/s", 78.6))
    if under.get("ms_per_step") and under.get("whole_path_tflops"):
        gf = under["whole_path_tflops"] * under["ms_per_step"]  # TFLOP/s x ms = GFLOP per step (all stages, all units)
        union_rows.append(("_whole_step:bench line of this run (under rocprofv3)", "", under["ms_per_step"] * under.get("steps", 1),
                           under.get("steps", 1), under["ms_per_step"], gf, "GFLOP", under["whole_path_tflops"], "TFLOP/s", 78.6))
    fiv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in tr if "k_fill_dense" in r["Kernel_Name"]]
    fill = (under.get("fill") or {})
    if fiv and fill.get("algorithmic_bytes_per_launch"):
        nfill = max(1, sum(1 for r in tr if "k_fill_dense_plain" in r["Kernel_Name"]))
        uni = interval_union(fiv)
        gb = fill["algorithmic_bytes_per_launch"] / 1e9
        ms_step = uni / 1e6 / nfill
        union_rows.append(("_union:dense fill (k_fill_dense_plain + k_fill_dense_band, two streams)", len(fiv), uni / 1e6, nfill,
                           ms_step, gb, "GB", gb / ms_step * 1e3, "GB/s", 8000.0))
with open(dst + "_kernel_stats.csv", "w") as fh:
    fh.write("kernel,calls,total_ms,avg_us,percent,steps,ms_per_step,algorithmic_per_step,algorithmic_unit,rate,rate_unit,frac_of_peak\n")
    for r in stats:
        fh.write(f"{short(r['Name'])},{r['Calls']},{int(r['TotalDurationNs'])/1e6:.3f},"
                 f"{float(r['AverageNs'])/1e3:.2f},{float(r['Percentage']):.2f},,,,,,,\n")
    # (total_ms of a `_union:` row = elapsed time covered by the family's launches, overlaps counted once)
    for name, calls_u, tot, steps_u, ms_step, alg, unit, rate, runit, peak in union_rows:
        fh.write(f"{name},{calls_u},{tot:.3f},,,{steps_u},{ms_step:.4f},{'' if alg is None else f'{alg:.3f}'},{unit},"
                 f"{'' if rate is None else f'{rate:.3f}'},{runit},{'' if rate is None else f'{rate / peak:.4f}'}\n")

agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
dur = collections.defaultdict(float)
sqdur = collections.defaultdict(float)
for d in sorted(os.listdir(src)):
    f = find(os.path.join(d, "**", "*counter_collection.csv"))
    if not f:
        continue
    rows = list(csv.DictReader(open(f)))
    seen = set()
    for r in rows:
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
# shader clock for the pipe-busy fraction: what the plain bench run of the same script measured (boxes hold 2.1 - 2.4 GHz)
clock_hz = 2.25e9
try:
    with open(os.path.join(src, "bench_plain.json")) as fh:
        line = [ln for ln in fh if ln.startswith("{")][-1]
    mhz = (json.loads(line).get("roofline") or {}).get("sustained_clock_mhz")
    if mhz and 1000 < mhz < 3000:
        clock_hz = mhz * 1e6
except Exception:
    pass
summary = {}
for k, v in agg.items():
    e = {"launches_per_step": calls[k] / PMC_STEPS, "ms_per_step_under_pmc": round(dur[k] / PMC_STEPS, 3)}
    if "FETCH_SIZE" in v:
        e["fetch_GB_per_step_corrected_x2"] = round(2 * v["FETCH_SIZE"] * 1024 / 1e9 / PMC_STEPS, 3)
    if "WRITE_SIZE" in v:
        e["write_GB_per_step"] = round(v["WRITE_SIZE"] * 1024 / 1e9 / PMC_STEPS, 3)
    if "TCC_HIT_sum" in v:
        e["l2_hit_rate"] = round(v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 3)
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES") and sqdur[k] > 0:
        # busy cycles summed over the 1024 SIMDs / (SIMDs x kernel time x the sustained clock of the plain run)
        e["mfma_pipe_busy_frac"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * sqdur[k] * 1e-3 * clock_hz), 3)
    if "fetch_GB_per_step_corrected_x2" in e or "write_GB_per_step" in e:
        e["hbm_bytes_per_launch"] = (e.get("fetch_GB_per_step_corrected_x2", 0) + e.get("write_GB_per_step", 0)) * 1e9 / max(1.0, e["launches_per_step"])
    summary[k] = e
g = [k for k in summary if any(k.startswith(p) for p in PANEL)]
tot_fetch = sum(summary[k].get("fetch_GB_per_step_corrected_x2", 0) for k in g)
tot_write = sum(summary[k].get("write_GB_per_step", 0) for k in g)
tot_launch = sum(summary[k]["launches_per_step"] for k in g)
summary["_k_chol_panel_all"] = {
    "kernels": g,
    "launches_per_step": tot_launch,
    "hbm_GB_per_step": round(tot_fetch + tot_write, 3),
    "hbm_bytes_per_launch": (tot_fetch + tot_write) * 1e9 / max(1, tot_launch),
}
if union:
    summary["_k_chol_panel_all"]["kernel_trace"] = union
try:  # the id of the code the counters were collected on (bench.py reports the same id in its line)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench

    summary["_code_id"] = bench.code_id()
    summary["_clock_mhz_for_pipe_busy"] = round(clock_hz / 1e6, 1)
except Exception as e:  # pragma: no cover
    summary["_code_id"] = None
json.dump(summary, open(dst + "_pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(summary["_k_chol_panel_all"]))
for k in sorted(summary):
    print(k, summary[k])
