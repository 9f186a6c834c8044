cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_f; mkdir -p $O; rm -f $O/potrf_times.txt
(timeout 600 python -m pytest tests/test_gpu_stages.py -m gpu -x -q -k "potrf and dataflow") > $O/tests_potrf.log 2>&1; tail -3 $O/tests_potrf.log
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
for b in ${BATCHES:-16 32 64 128}; do for seq in ${SEQS:-0 4}; do echo "B=$b seq=$seq" >> $O/potrf_times.txt; timeout 120 python tools/bench_potrf.py ${NN:-4096} $b 3 $seq 2>&1 | grep -E "potrf|max|clock during" >> $O/potrf_times.txt; done; done
grep -E "^B=|potrf [0-9]|max" $O/potrf_times.txt | awk '/^B=/{h=$0} /potrf [0-9]/{print h, $4, $5, $6,$7,$8} /max/{print h, $0}'
