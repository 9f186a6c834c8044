cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_d; mkdir -p $O
(time python -m pytest tests/test_gpu_model.py tests/test_gpu_api.py tests/test_gpu_stages.py -m gpu -x -q -k "cov_fill or non_positive or poisoned or forward or small_model or cfg2 or api or call") > $O/tests_new.log 2>&1; tail -5 $O/tests_new.log
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
for v in "SF_FILL_OLD=1" "SF_FILL_TWO_STREAMS=1" "SF_FILL_TWO_STREAMS=0" "SF_FILL_SPAN4=1"; do
  env $v python tools/bench_fill.py 4096 128 4096 2>/dev/null | tail -1 >> $O/fill_variants.jsonl
done
env SF_FILL_TWO_STREAMS=1 python tools/bench_fill.py 4096 128 4112 2>/dev/null | tail -1 >> $O/fill_variants.jsonl
env SF_FILL_TWO_STREAMS=1 python tools/bench_fill.py 3000 64 3000 2>/dev/null | tail -1 >> $O/fill_variants.jsonl
env SF_FILL_OLD=1 python tools/bench_fill.py 3000 64 3000 2>/dev/null | tail -1 >> $O/fill_variants.jsonl
cat $O/fill_variants.jsonl
