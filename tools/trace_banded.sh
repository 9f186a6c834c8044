# Per-launch timeline of one banded step (tuning aid):  [SF_BENCH_LS=..] bash tools/trace_banded.sh [N] [B]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/trace_banded; mkdir -p $R/gpurun_out
SF_COMPARE_DENSE=0 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_banded -- python $R/tools/bench_banded.py ${1:-4096} ${2:-128} 1 > $R/gpurun_out/trace_banded.log 2>&1
grep "banded:" $R/gpurun_out/trace_banded.log
python - <<'PY'
import csv, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
f = glob.glob(R + "/gpurun_out/trace_banded/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last step = everything after the last k_band_fill launch
last = max(i for i, r in enumerate(rows) if "k_band_fill" in r["Kernel_Name"])
sel = rows[max(0, last - 3):]
t0 = int(sel[0]["Start_Timestamp"])
print("start_us  dur_us  end_us  queue grid  kernel")
for r in sel[:140]:
    st, en = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{st/1e3:9.1f} {(en-st)/1e3:8.1f} {en/1e3:9.1f}  q{r.get('Queue_Id','?')} {r.get('Grid_Size','?'):>8} {r['Kernel_Name'].replace('void ','')[:40]}")
PY
