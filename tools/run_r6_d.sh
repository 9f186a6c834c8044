#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6d; mkdir -p $O
NAME=r6d/ab_order TAGS="exp base" ROUNDS=3 CASES="4096 32 5 4;4096 64 4 4;4096 48 4 4" bash tools/run_ab.sh > /dev/null 2>&1
export SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_exp.so
K=$O/knobs.txt; : > $K
for r in 1 2; do
for c in "4096 3 6" "4096 5 6" "4096 6 6" "4096 7 6" "4096 9 5" "4096 12 5" "4096 15 5" "4096 20 4" "4096 28 4" "4096 50 3" "8192 12 2" "8192 5 3" "2048 12 8" "2048 100 4"; do
  for kn in "SF_DF_QBAL=0" "SF_DF_QBAL=1"; do
    echo "== $c | $kn" >> $K
    env $kn timeout 300 python tools/bench_potrf.py $c 4 2>&1 | grep -E "potrf [0-9]|max .L" >> $K
  done
done
done
for c in "4096 12 5" "4096 50 3" "8192 12 2"; do
    echo "== $c | seq 0" >> $K
    timeout 300 python tools/bench_potrf.py $c 0 2>&1 | grep -E "potrf [0-9]" >> $K
done
(time timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_recovery.py tests/test_gpu_train.py -m gpu -q -x) > $O/tests.log 2>&1; tail -4 $O/tests.log
