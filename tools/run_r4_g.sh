cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_g; mkdir -p $O
export SF_LIB_PATH=$R/starfish_amd/libstarfish_amd_tuning.so
for seq in 0 4; do for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  name=$(echo $set | cut -d' ' -f1)
  rm -rf $O/pmc_$seq_$name
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${seq}_$name -o pmc -- python $R/tools/bench_potrf.py 4096 ${BB:-128} 1 $seq > $O/log_${seq}_$name.txt 2>&1
done; done
python - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04_g"
for seq in (0,4):
    agg=collections.defaultdict(float)
    for d in glob.glob(f"{O}/pmc_{seq}_*"):
        for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k=r["Kernel_Name"]
                if "k_chol_panel" in k or "k_potrf_dataflow" in k or "k_diag" in k:
                    agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(seq, {k: f"{v:.4g}" for k,v in agg.items()}, "hit", agg["TCC_HIT_sum"]/(agg["TCC_HIT_sum"]+agg["TCC_MISS_sum"]+1), "fetch GB x2", 2*agg["FETCH_SIZE"]*1024/1e9)
PY
