#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6m; mkdir -p $O
export SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_tuning.so
K=$O/knobs.txt; : > $K
for r in 1 2; do
for c in "4096 16 5" "4096 32 4" "4096 64 3"; do
  for kn in "SF_DF_CAP=0" "SF_DF_CAP=96" "SF_DF_CAP=128" "SF_DF_CAP=48" "SF_DF_FRONT=1" "SF_DF_FRONT=2" "SF_DF_FRONT=3" "SF_DF_FRONT=4" "SF_DF_PT_TASKS=8" "SF_DF_PT_TASKS=32" "SF_DF_FP_POS=0" "SF_DF_FP_POS=128" "SF_DF_TAIL=0" "SF_DF_TAIL=8" "SF_DF_TAIL=16"; do
    echo "== $c | $kn" >> $K
    env $kn timeout 300 python tools/bench_potrf.py $c 4 2>&1 | grep -E "potrf [0-9]" >> $K
  done
done
done
tail -3 $K
