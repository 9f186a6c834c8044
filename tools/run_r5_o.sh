cd $GRAFT_REPO_ROOT
O=gpurun_out/r5o; rm -rf $O; mkdir -p $O
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
SF_DIAG_STAMPS=1 timeout 200 python tools/bench_potrf.py 1024 1 1 0 2>&1 | grep k_diag | tail -2 | tee $O/diag_stamps.txt
for rep in 1 2 3; do for t in 1 2 4; do
  echo "tpw=$t potrf B=128 $(SF_WIDE_TPW=$t timeout 120 python tools/bench_potrf.py 4096 128 3 2 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" | tee -a $O/ab.txt
  echo "tpw=$t bench $(SF_WIDE_TPW=$t python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-structured --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")" | tee -a $O/ab.txt
done; done
for t in 1 2 4; do
  echo "tpw=$t cfg3 $(SF_WIDE_TPW=$t python bench.py --config cfg3 --steps 3 --warmup 1 --cpu-sample 0 --no-structured --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")" | tee -a $O/ab.txt
done
