#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; mkdir -p $O
NAME=r6c/ab_prog TAGS="base exp noprog" ROUNDS=3 CASES="4096 32 5 4;4096 64 4 4;4096 16 5 4;4096 48 4 4" bash tools/run_ab.sh > /dev/null 2>&1
export SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_exp.so
K=$O/knobs.txt; : > $K
for r in 1 2; do
for c in "4096 1 8" "4096 3 6" "4096 5 6" "4096 6 6" "4096 7 6" "4096 2 6" "4096 4 6" "2048 4 8" "1024 4 8" "8192 2 3" "16384 1 2" "16384 2 2"; do
  for kn in "SF_DF_QBAL=0" "SF_DF_QBAL=1" "SF_DF_QBAL=1 SF_DF_CAP=64" "SF_DF_QBAL=1 SF_DF_CAP=256"; do
    echo "== $c | $kn" >> $K
    env $kn timeout 300 python tools/bench_potrf.py $c 4 2>&1 | grep -E "potrf [0-9]" >> $K
  done
done
done
for c in "8192 8 2" "8192 16 2" "8192 12 2" "16384 2 2" "16384 1 2" "8192 2 3" "8192 1 3"; do
  for seq in 4 0; do
    echo "== $c | seq $seq" >> $K
    timeout 300 python tools/bench_potrf.py $c $seq 2>&1 | grep -E "potrf [0-9]" >> $K
  done
done
tail -5 $K
