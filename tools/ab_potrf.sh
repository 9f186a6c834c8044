# same-box A/B of library builds on sf_potrf_batch (run on the GPU box; builds: make TAG=<name> in starfish_amd/csrc, e.g. from a stash of
# the previous commit):  LIBS="base tuning" BATCHES="16 32 64" [KN="SF_X=1 SF_Y=2"] [NN=4096] [SEQ=4] bash tools/ab_potrf.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab2.txt; rm -f $O
for rep in 1 2; do for b in ${BATCHES:-32 64}; do for l in ${LIBS:-base tuning noseg}; do
  export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_$l.so
  echo "lib=$l B=$b $(env ${KN:-X=0} timeout 120 python tools/bench_potrf.py ${NN:-4096} $b 3 ${SEQ:-4} 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" >> $O
done; done; done
sort $O
