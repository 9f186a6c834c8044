# HBM / L2 counters of one sf_potrf_batch call per launch sequence (tuning aid):  bash tools/pmc_potrf.sh <tag> [N] [B] [seqs]
# separate --pmc passes (never combined with runtime traces); sums per kernel family over the whole run (warm-up + 1 rep)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; N=${2:-4096}; B=${3:-128}; SEQS=${4:-"0 2"}
OUT=$R/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
for seq in $SEQS; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=seq${seq}_$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$name -o pmc -- python $R/tools/bench_potrf.py $N $B 1 $seq > $OUT/$name.log 2>&1 || tail -3 $OUT/$name.log
  done
done
python - <<PY
import csv, glob, collections, json
out={}
for d in sorted(glob.glob("$OUT/seq*_*")):
    if not d.endswith(("SIZE","sum")): continue
    f=glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not f: continue
    acc=collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        fam="panel_w" if "k_chol_panel_w" in k else "panel" if "k_chol_panel" in k else "diag" if "k_diag" in k else "gemm_nt" if "k_gemm_nt" in k else None
        if fam: acc[(fam,r["Counter_Name"])]+=float(r["Counter_Value"])
    key=d.split("/")[-1].split("_")[0]
    for (fam,c),v in acc.items(): out.setdefault(key,{}).setdefault(fam,{})[c]=v
# two potrf calls per run (warm-up + 1 timed): per call = /2 ; FETCH_SIZE in KB, x2 (gfx950 correction of the micro-arch guide)
res={}
for key,fams in out.items():
    res[key]={}
    for fam,c in fams.items():
        e={}
        if "FETCH_SIZE" in c: e["hbm_read_GB_per_call"]=c["FETCH_SIZE"]*2*1024/2/1e9
        if "WRITE_SIZE" in c: e["hbm_write_GB_per_call"]=c["WRITE_SIZE"]*1024/2/1e9
        if "TCC_HIT_sum" in c: e["l2_hit_rate"]=c["TCC_HIT_sum"]/max(1.0,c["TCC_HIT_sum"]+c["TCC_MISS_sum"])
        res[key][fam]=e
json.dump(res, open("$OUT.json","w"), indent=1)
print(json.dumps(res, indent=1))
PY
