cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n; rm -rf $O; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_banded.py -x -q) > $O/tests.log 2>&1; tail -2 $O/tests.log
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
SF_DIAG_STAMPS=1 timeout 200 python tools/bench_potrf.py 1024 1 1 0 2>&1 | grep k_diag | tee $O/diag_stamps.txt
for b in 8 16 32 64; do echo "B=$b $(timeout 120 python tools/bench_potrf.py 4096 $b 3 4 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" | tee -a $O/df.txt; done
