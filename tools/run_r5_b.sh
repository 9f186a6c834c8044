# round 5: dynamic FP(k,1) claims -- parity under the dataflow sequence, A/B through the tuning build's switch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; rm -rf $O; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_gpu_recovery.py tests/test_gpu_stages.py -x -q) > $O/tests.log 2>&1; tail -5 $O/tests.log
for rep in 1 2; do for b in 4 8 16 24 32 48 64; do for v in "tuning 0" "tuning 1" "inl 1"; do set -- $v
  export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_$1.so
  echo "lib=$1 dyn=$2 B=$b $(SF_DF_FP1_DYN=$2 timeout 120 python tools/bench_potrf.py 4096 $b 3 4 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" >> $O/ab.txt
done; done; done
sort $O/ab.txt
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
for nb in "3008 16" "2048 16" "2048 64"; do for dyn in 0 1; do
  echo "dyn=$dyn N,B=$nb $(SF_DF_FP1_DYN=$dyn timeout 120 python tools/bench_potrf.py $nb 3 4 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" | tee -a $O/ab2.txt
done; done
SF_DF_VERBOSE=1 SF_DF_TRACE=1 timeout 120 python tools/bench_potrf.py 4096 16 1 4 > $O/trace_b16.txt 2>&1
