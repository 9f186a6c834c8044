#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6k; mkdir -p $O
F=$O/cfg3_ab.txt; : > $F
for r in 1 2 3; do for t in exp call; do
  echo "== $t cfg3" >> $F
  SF_ALLOW_OLD_LIB=1 SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_$t.so timeout 600 python bench.py --config cfg3 --steps 4 --warmup 1 --cpu-sample 0 --no-structured --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_ms_per_step'])" >> $F
done; done
cat $F
