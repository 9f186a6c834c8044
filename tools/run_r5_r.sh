cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r; rm -rf $O; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k "line_phases or cov_fill") > $O/tests.log 2>&1; tail -15 $O/tests.log | cut -c1-200
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
for ns in 0 1; do echo "noshift=$ns N=3000 $(SF_FILL_NO_SHIFT=$ns python tools/bench_fill.py 3000 128 3000 2>/dev/null | tail -1)" | tee -a $O/fill.txt; done
echo "N=3000 ld=3008 $(python tools/bench_fill.py 3000 128 3008 2>/dev/null | tail -1)" | tee -a $O/fill.txt
