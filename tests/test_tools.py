"""tools/summarize_profile.py on a hand-made rocprofv3 output directory: the committed *_kernel_stats.csv must be enough to
recompute a roofline fraction -- per-kernel totals of launches that overlap on several streams are not."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_kernel_stats_csv_carries_union_rows_that_give_the_roofline_fraction(tmp_path):
    src = tmp_path / "gpurun_out" / "tag"
    (src / "stats" / "host").mkdir(parents=True)
    # two steps; per step three panel launches, two of them side by side on two streams, and one k_finish
    trace = [("k_chol_panel_w<true>", 0, 100), ("k_chol_panel<true,0>", 50, 150), ("k_potrf_dataflow<true>", 200, 260), ("k_finish", 260, 262),
             ("k_chol_panel_w<true>", 1000, 1100), ("k_chol_panel<true,0>", 1050, 1150), ("k_potrf_dataflow<true>", 1200, 1260),
             ("k_finish", 1260, 1262), ("k_fill_tiles", 900, 950)]
    with open(src / "stats" / "host" / "bench_kernel_trace.csv", "w") as fh:
        fh.write("Kernel_Name,Start_Timestamp,End_Timestamp\n")
        for name, a, b in trace:
            fh.write(f'"{name}",{a * 1000000},{b * 1000000}\n')  # (ms -> ns)
    with open(src / "stats" / "host" / "bench_kernel_stats.csv", "w") as fh:
        fh.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        fh.write('"k_chol_panel_w<true>",2,200000000,100000000,40.0\n"k_chol_panel<true,0>",2,200000000,100000000,40.0\n')
    # the bench line of the same run: 1 profiled step, 3 launches of 7e12 / 3 algorithmic flops
    line = {"value": 10.0, "ms_per_step": 262.0, "steps": 2, "whole_path_tflops": 30.0,
            "roofline": {"algorithmic_flops_per_launch": 7e12 / 3, "launches": 3, "profiled_steps": 1}}
    (src / "bench_under_rocprof.json").write_text("noise\n" + json.dumps(line) + "\n")
    dst = tmp_path / "profiles_tag"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_profile.py"), str(src), str(dst)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = list(csv.DictReader(open(str(dst) + "_kernel_stats.csv")))
    by = {r["kernel"].split(":")[0]: r for r in rows}
    u = by["_union"]
    # union per step: [0, 150] + [200, 260] = 210 ms (the plain sum is 260); 7000 GFLOP / 210 ms = 33.33 TFLOP/s = 0.424 of 78.6
    assert int(u["calls"]) == 6 and int(u["steps"]) == 2
    assert abs(float(u["total_ms"]) - 420.0) < 1e-6 and abs(float(u["ms_per_step"]) - 210.0) < 1e-6
    assert abs(float(u["algorithmic_per_step"]) - 7000.0) < 1e-6 and u["algorithmic_unit"] == "GFLOP"
    assert abs(float(u["rate"]) - 7000.0 / 210.0) < 1e-3 and abs(float(u["frac_of_peak"]) - 7000.0 / 210.0 / 78.6) < 1e-4
    w = by["_whole_step"]
    assert abs(float(w["ms_per_step"]) - 262.0) < 1e-9 and abs(float(w["rate"]) - 30.0) < 1e-9
    assert abs(float(w["frac_of_peak"]) - 30.0 / 78.6) < 1e-4
    # the per-kernel rows are still there, and the JSON summary still carries the union
    assert by["k_chol_panel_w<true>"]["calls"] == "2"
    summ = json.load(open(str(dst) + "_pmc_summary.json"))
    assert abs(summ["_k_chol_panel_all"]["kernel_trace"]["union_ms_total"] - 420.0) < 1e-6
