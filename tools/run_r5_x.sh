cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SF_LIB_PATH=$R/starfish_amd/libstarfish_amd_tuning.so
O=$R/gpurun_out/r5x; rm -rf $O; mkdir -p $O
for skip in 0 1 2 3 4 7; do
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $O/s$skip -o pmc -- env SF_WIDE_SKIP=$skip python $R/tools/bench_potrf.py 4096 128 1 2 > $O/s$skip.log 2>&1
  python - $O/s$skip $skip <<'P'
import csv,glob,sys,collections
acc=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_chol_panel_w" in r["Kernel_Name"]: acc[r["Counter_Name"]]+=float(r["Counter_Value"])
print("skip", sys.argv[2], dict(acc), "ratio", acc["SQ_LDS_BANK_CONFLICT"]/max(1,acc["SQ_ACTIVE_INST_LDS"]))
P
  grep "potrf " $O/s$skip.log | tail -1
done
