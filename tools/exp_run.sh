cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3s
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_model.py -m gpu -x -q -k "wide" > gpurun_out/r3s/t1.log 2>&1; tail -3 gpurun_out/r3s/t1.log
export SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_tuning.so
for tr in -1 2 5 8 12; do
  echo "tail_rounds=$tr: $(SF_WIDE_TAIL_ROUNDS=$tr python tools/bench_potrf.py 4096 128 3 2>&1 | grep "^N=" | tail -2 | tr '\n' ' ')"
done
echo "narrow: $(SF_CHOL_UNFUSED=0 python tools/bench_potrf.py 4096 128 3 2>&1 | grep "^N=" | tail -1)"
