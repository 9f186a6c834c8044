# SQ counters of one sf_potrf_batch call (tuning aid): LDS bank conflicts, LDS / VMEM wait share, instruction mix per kernel family
#   bash tools/pmc_sq_potrf.sh <tag> [N] [B] [seq]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; N=${2:-4096}; B=${3:-128}; SEQ=${4:-2}
OUT=$R/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
for set in "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_INSTS_SALU"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$name -o pmc -- python $R/tools/bench_potrf.py $N $B 1 $SEQ > $OUT/$name.log 2>&1 || tail -3 $OUT/$name.log
done
python - <<PY
import csv, glob, collections, json
acc=collections.defaultdict(float)
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        fam="panel_w" if "k_chol_panel_w" in k else "panel" if "k_chol_panel" in k else "diag" if "k_diag" in k else "dataflow" if "k_potrf_dataflow" in k else None
        if fam: acc[(fam,r["Counter_Name"])]+=float(r["Counter_Value"])
res=collections.defaultdict(dict)
for (fam,c),v in acc.items(): res[fam][c]=v
for fam,c in res.items():
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_ACTIVE_INST_LDS"): c["lds_conflict_cycles_per_active_lds_cycle"]=c["SQ_LDS_BANK_CONFLICT"]/c["SQ_ACTIVE_INST_LDS"]
    if "SQ_WAIT_INST_ANY" in c and c.get("SQ_WAVE_CYCLES"): c["wait_inst_any_frac_of_wave_cycles"]=c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"]
json.dump(res, open("$OUT.json","w"), indent=1)
print(json.dumps(res, indent=1))
PY
