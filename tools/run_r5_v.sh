cd $GRAFT_REPO_ROOT
O=gpurun_out/r5v; rm -rf $O; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_stages.py -x -q -k "wide or potrf or spd") > $O/tests.log 2>&1; tail -2 $O/tests.log
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
SF_WIDE_STAMPS=1 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-structured --no-extra-legs > $O/stamps_bench.txt 2>&1
grep -A45 "wide launches" $O/stamps_bench.txt | tail -46 | awk 'NR%4==1'
for rep in 1 2 3; do echo "bench $(python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-structured --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")"; done
unset SF_LIB_PATH
for rep in 1 2 3; do echo "release(old) bench $(python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-structured --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")"; done
