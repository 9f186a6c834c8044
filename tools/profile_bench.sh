# rocprofv3 evidence for one bench configuration (run on the GPU box):
#   bash tools/profile_bench.sh <tag> [bench.py args...]      e.g.  bash tools/profile_bench.sh r02_cfg2 --config cfg2
# 1. kernel trace + stats of the default-length run  -> gpurun_out/<tag>/stats
# 2. four separate --pmc passes of `--steps 1 --warmup 1` (two identical steps)  -> gpurun_out/<tag>/<COUNTER>
# (counters are never combined with the runtime/sys trace domains); then tools/summarize_profile.py writes
# profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc_summary.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py "$@" --cpu-sample 0 --no-structured --no-extra-legs > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$name -o pmc -- python $R/bench.py "$@" --steps 1 --warmup 1 --cpu-sample 0 --no-structured --no-extra-legs > $OUT/$name.log 2>&1 || tail -3 $OUT/$name.log
done
python $R/bench.py "$@" --cpu-sample 0 --no-structured --no-extra-legs > $OUT/bench_plain.json 2>/dev/null
python $R/tools/summarize_profile.py $OUT $R/gpurun_out/$TAG
