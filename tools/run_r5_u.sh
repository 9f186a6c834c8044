cd $GRAFT_REPO_ROOT
O=gpurun_out/r5u; rm -rf $O; mkdir -p $O
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
for rep in 1 2; do for b in 8 16 24 32 64; do for pt in 4 8 16 32; do
  echo "B=$b pt=$pt $(SF_DF_PT_TASKS=$pt timeout 120 python tools/bench_potrf.py 4096 $b 3 4 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" >> $O/pt.txt
done; done; done
sort $O/pt.txt
