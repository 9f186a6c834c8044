#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6i; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_recovery.py -m gpu -q -x) > $O/tests.log 2>&1; tail -3 $O/tests.log
export SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_exp.so
K=$O/knobs.txt; : > $K
for r in 1 2 3; do
for c in "4096 16 5" "4096 32 5" "4096 64 4" "4096 8 5" "4096 1 8" "4096 24 4" "2048 16 8" "3008 32 5" "1024 32 10"; do
  for kn in "SF_DF_HINT=0" "SF_DF_HINT=1"; do
    echo "== $c | $kn" >> $K
    env $kn timeout 300 python tools/bench_potrf.py $c 4 2>&1 | grep -E "potrf [0-9]" >> $K
  done
done
done
unset SF_LIB_PATH
timeout 900 python tools/stress_potrf.py > $O/stress.txt 2>&1; tail -6 $O/stress.txt
