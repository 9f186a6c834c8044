#!/bin/bash
# soak of the persistent kernel (sf_potrf_batch pinned to sequence 4, tools/stress_potrf.py) over many shapes:  bash tools/soak_potrf.sh > gpurun_out/soak.txt
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
for c in "4096 1 2000" "4096 2 1000" "4096 3 1000" "4096 5 1000" "4096 12 1000" "4096 16 2000" "4096 28 600" "4096 64 400" "2048 32 2000" "2048 7 1500" "1024 24 3000" "3008 8 1500" "3008 33 600" "512 100 2000" "8192 5 200" "8192 2 300" "16384 2 60" "16384 9 20"; do
  timeout 900 python tools/stress_potrf.py $c 4 2>&1 | tail -1
done
