# round 5: all front partial sums claimed when ready -- A/B through the tuning build's switch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; rm -rf $O; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_gpu_recovery.py tests/test_gpu_stages.py -x -q) > $O/tests.log 2>&1; tail -3 $O/tests.log
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
for rep in 1 2; do for b in 4 8 16 24 32 48 64; do for dyn in 0 1; do
  echo "dyn=$dyn B=$b $(SF_DF_FP1_DYN=$dyn timeout 120 python tools/bench_potrf.py 4096 $b 3 4 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" >> $O/ab.txt
done; done; done
sort $O/ab.txt
for nb in "3008 16" "2048 16" "2048 64" "1024 16"; do for dyn in 0 1; do
  echo "dyn=$dyn N,B=$nb $(SF_DF_FP1_DYN=$dyn timeout 120 python tools/bench_potrf.py $nb 3 4 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" | tee -a $O/ab2.txt
done; done
SF_DF_VERBOSE=1 SF_DF_TRACE=1 timeout 120 python tools/bench_potrf.py 4096 16 1 4 > $O/trace_b16.txt 2>&1
