#!/bin/bash
# same-box A/B of experiment builds:  TAGS="base glate" bash tools/run_ab.sh  (libs: starfish_amd/libstarfish_amd_<tag>.so)
# CASES: "N B reps seq" separated by ';' (seq as tools/bench_potrf.py: 0 fused, 2 wide, 4 dataflow, empty = automatic)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/${NAME:-ab}.txt
: > $OUT
TAGS=${TAGS:-"base exp"}
CASES=${CASES:-"4096 128 3 2;4096 128 3 0;4096 64 4;4096 32 4;4096 16 4"}
ROUNDS=${ROUNDS:-3}
IFS=';' read -ra CS <<< "$CASES"
for r in $(seq 1 $ROUNDS); do
  for c in "${CS[@]}"; do
    for t in $TAGS; do
      echo "== round $r tag $t case $c" >> $OUT
      SF_ALLOW_OLD_LIB=1 SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_$t.so timeout 300 python tools/bench_potrf.py $c 2>&1 | grep -E "potrf [0-9]|during|max .L" >> $OUT
    done
  done
done
tail -n 120 $OUT
