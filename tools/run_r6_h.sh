#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h; mkdir -p $O
F=$O/fill_ab.txt; : > $F
for r in 1 2 3; do
for t in inl exp; do
  for c in "4096 128 4096" "3000 128 3000" "3000 128 3008"; do
    for occ in 2 3; do
      echo "== $t | $c | occ $occ" >> $F
      SF_FILL_BAND_OCC=$occ SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_$t.so timeout 300 python tools/bench_fill.py $c 2>/dev/null | tail -1 >> $F
    done
  done
done
done
# the likelihood's own fill (tile list) + whole step, both builds: bench stage times
for t in inl exp; do
  echo "== $t bench cfg2" >> $F
  SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_$t.so timeout 600 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-structured --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_ms_per_step'])" >> $F
done
(time timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_model.py tests/test_gpu_api.py -m gpu -q -x) > $O/tests.log 2>&1; tail -3 $O/tests.log
tail -n 8 $F
