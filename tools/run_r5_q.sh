cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q; rm -rf $O; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_model.py tests/test_gpu_fullbatch.py -x -q -k "wide or potrf or spd or cfg") > $O/tests.log 2>&1; tail -3 $O/tests.log
for rep in 1 2 3; do for l in base tuning; do
  export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_$l.so
  echo "lib=$l potrf B=128 $(timeout 120 python tools/bench_potrf.py 4096 128 3 2 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" | tee -a $O/ab.txt
  echo "lib=$l bench $(python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-structured --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")" | tee -a $O/ab.txt
done; done
for l in base tuning; do
  export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_$l.so
  echo "lib=$l cfg3 $(python bench.py --config cfg3 --steps 3 --warmup 1 --cpu-sample 0 --no-structured --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")" | tee -a $O/ab.txt
done
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
SF_WIDE_STAMPS=1 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-structured --no-extra-legs > $O/stamps_bench.txt 2>&1
grep -A45 "wide launches" $O/stamps_bench.txt | tail -46 | awk 'NR%3==1'
