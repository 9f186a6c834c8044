cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g; rm -rf $O; mkdir -p $O
(timeout 1500 python -m pytest tests/test_bench_launcher.py -x -q -m gpu -k eight) > $O/t8.log 2>&1; tail -3 $O/t8.log; grep -n "^E " $O/t8.log | cut -c1-1800 | head -6
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
for rep in 1 2; do for h in 0 2 4 6 8 12; do
  echo "head=$h B=128 $(SF_WIDE_HEAD=$h timeout 120 python tools/bench_potrf.py 4096 128 3 2 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" | tee -a $O/head.txt
done; done
for h in 0 4 8; do echo "head=$h bench $(SF_WIDE_HEAD=$h python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-structured --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")" | tee -a $O/head.txt; done
for h in 0 4 8; do echo "head=$h cfg3 $(SF_WIDE_HEAD=$h python bench.py --config cfg3 --steps 3 --warmup 1 --cpu-sample 0 --no-structured --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")" | tee -a $O/head.txt; done
NN=4096 BB=128 SEQ=0 bash tools/trace_potrf.sh > $O/timeline_b128_fused.txt 2>&1
