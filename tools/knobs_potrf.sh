# same-box comparison of tuning-build environment settings on sf_potrf_batch (run on the GPU box):
#   KNOBS="SF_DF_TAIL=0|SF_DF_TAIL=6|SF_DF_TAIL=6 SF_DF_TAIL_FRONT=2" BATCHES="16 32" [NN=4096] [SEQ=4] bash tools/knobs_potrf.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/knobs.txt; rm -f $O
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
IFS='|' read -ra SETS <<< "${KNOBS:-X=0}"
for rep in 1 2; do for b in ${BATCHES:-32 64}; do for ks in "${SETS[@]}"; do
  echo "B=$b [$ks] $(env $ks timeout 120 python tools/bench_potrf.py ${NN:-4096} $b 3 ${SEQ:-4} 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" >> $O
done; done; done
sort $O
