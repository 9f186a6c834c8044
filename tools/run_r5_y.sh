cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5y; rm -rf $O; mkdir -p $O
for l in tuning t129 t131 t133 t134 t137 t138 t142; do
  export SF_LIB_PATH=$R/starfish_amd/libstarfish_amd_$l.so
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $O/$l -o pmc -- python $R/tools/bench_potrf.py 4096 128 1 2 > $O/$l.log 2>&1
  python - $O/$l $l <<'P'
import csv,glob,sys,collections
acc=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_chol_panel_w" in r["Kernel_Name"]: acc[r["Counter_Name"]]+=float(r["Counter_Value"])
print("lib", sys.argv[2], "conflict/active", acc["SQ_LDS_BANK_CONFLICT"]/max(1,acc["SQ_ACTIVE_INST_LDS"]), acc["SQ_LDS_BANK_CONFLICT"])
P
done
cd $R
for rep in 1 2 3; do for l in tuning t129 t131 t133 t134 t137 t138 t142; do
  export SF_LIB_PATH=$R/starfish_amd/libstarfish_amd_$l.so
  echo "lib=$l potrf $(timeout 120 python tools/bench_potrf.py 4096 128 3 2 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" | tee -a $O/ab.txt
done; done
for l in t129 t133; do
export SF_LIB_PATH=$R/starfish_amd/libstarfish_amd_$l.so
SF_WIDE_STAMPS=1 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-structured --no-extra-legs > $O/stamps_$l.txt 2>&1
echo $l; grep -A45 "wide launches" $O/stamps_$l.txt | tail -46 | awk 'NR%6==2'
done
