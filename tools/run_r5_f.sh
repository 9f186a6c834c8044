cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; rm -rf $O; mkdir -p $O
(timeout 1500 python -m pytest tests/test_bench_launcher.py -x -q -m gpu -k eight) > $O/t8.log 2>&1; tail -5 $O/t8.log; grep -n "^E " $O/t8.log | cut -c1-1500 | head -8
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
for occ in 2 3 4; do echo "occ=$occ $(SF_FILL_BAND_OCC=$occ python tools/bench_fill.py 2>/dev/null | tail -1)" | tee -a $O/fill.txt; done
for nl in "3000 3000" "3000 3008" "3008 3008" "2944 2944" "3072 3072"; do set -- $nl; echo "N=$1 ld=$2 $(python tools/bench_fill.py $1 128 $2 2>/dev/null | tail -1)" | tee -a $O/fill.txt; done
