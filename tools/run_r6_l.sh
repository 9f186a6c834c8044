#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6l; mkdir -p $O
SF_ALLOW_OLD_LIB=1 SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_exp.so timeout 1200 python -m pytest tests/test_gpu_stages.py tests/test_gpu_fullbatch.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
NAME=r6l/ab_wide TAGS="head exp" ROUNDS=3 CASES="4096 128 3 2;3008 512 2 2" bash tools/run_ab.sh > /dev/null 2>&1
NAME=r6l/ab_wide_rev TAGS="exp head" ROUNDS=2 CASES="4096 128 3 2" bash tools/run_ab.sh > /dev/null 2>&1
F=$O/bench_ab.txt; : > $F
for r in 1 2; do for t in head exp; do for cfg in cfg2 cfg3; do
  echo "== $t $cfg" >> $F
  SF_ALLOW_OLD_LIB=1 SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_$t.so timeout 600 python bench.py --config $cfg --steps 6 --warmup 2 --cpu-sample 0 --no-structured --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $F
done; done; done
cat $F
SF_WIDE_STAMPS=1 SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_exp.so timeout 300 python tools/bench_potrf.py 4096 128 1 2 2>&1 | grep -E "^ *(1[0-9]|[0-9]) +[0-9]+ \|" | head -20 > $O/stamps_exp.txt
SF_ALLOW_OLD_LIB=1 SF_WIDE_STAMPS=1 SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_head.so timeout 300 python tools/bench_potrf.py 4096 128 1 2 2>&1 | grep -E "^ *(1[0-9]|[0-9]) +[0-9]+ \|" | head -20 > $O/stamps_head.txt
head -12 $O/stamps_head.txt; echo; head -12 $O/stamps_exp.txt
