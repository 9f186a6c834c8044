cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
for v in v0 nobar prio fragpf v0; do
  echo "== $v" >> gpurun_out/r3c/exp1.log
  SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_$v.so python tools/bench_potrf.py 4096 128 3 >> gpurun_out/r3c/exp1.log 2>&1
done
cat gpurun_out/r3c/exp1.log | grep -v amdgpu.ids
(time python -m pytest tests -m gpu -x -q --durations=15) > gpurun_out/r3c/tests.log 2>&1; tail -25 gpurun_out/r3c/tests.log
