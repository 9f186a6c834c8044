# SQ counters of the banded sweep (tuning aid):  bash tools/pmc_band.sh   (separate --pmc passes, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pmcb; mkdir -p $R/gpurun_out/pmcb
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  SF_COMPARE_DENSE=0 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcb/p$i -- python $R/tools/bench_banded.py 4096 128 1 > $R/gpurun_out/pmcb/p$i.log 2>&1 || tail -3 $R/gpurun_out/pmcb/p$i.log
done
python - <<'PY'
import csv, glob, collections, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/pmcb/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_band_forms" in row["Kernel_Name"]:
            agg[row["Grid_Size"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for g, v in agg.items():
    print("k_band_forms grid", g)
    for c, vals in sorted(v.items()):
        print("   %-28s n=%d mean=%.4g" % (c, len(vals), sum(vals) / len(vals)))
PY
