# Per-launch timeline of one sf_potrf_batch call (tuning aid):  bash tools/trace_potrf.sh [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
NN=${NN:-4096}; BB=${BB:-128}
for kv in "$@"; do export "$kv"; done
# the environment switches exist only in the tuning build (make -C starfish_amd/csrc TUNING=1 on the authoring box)
[ -f $R/starfish_amd/libstarfish_amd_tuning.so ] && export SF_LIB_PATH=$R/starfish_amd/libstarfish_amd_tuning.so
rm -rf $R/gpurun_out/trace_potrf
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_potrf -- python $R/tools/bench_potrf.py $NN $BB 1 $SEQ > $R/gpurun_out/trace_potrf.log 2>&1
grep "potrf " $R/gpurun_out/trace_potrf.log
python - <<'PY'
import csv, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
f = glob.glob(R + "/gpurun_out/trace_potrf/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last potrf call = launches after the last elementwise copy kernel preceding the final k_gemm_nt burst
idx = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in ("k_gemm_nt", "k_diag_mfma", "k_chol_panel"))]
# split bursts by gaps > 2 ms
bursts, cur = [], [idx[0]]
for a, b in zip(idx, idx[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]) > 2_000_000:
        bursts.append(cur); cur = []
    cur.append(b)
bursts.append(cur)
# the timed repetition is the second burst of bench_potrf (first = warm-up); third = clock probe run
sel = bursts[1] if len(bursts) > 1 else bursts[0]
t0 = int(rows[sel[0]]["Start_Timestamp"])
print("start_us  dur_us  end_us  queue grid  kernel")
for i in sel:
    r = rows[i]
    st, en = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("void ", "")[:34]
    print(f"{st/1e3:9.1f} {(en-st)/1e3:8.1f} {en/1e3:9.1f}  q{r.get('Queue_Id','?')} {r.get('Grid_Size','?'):>8} {name}")
PY
