cd $GRAFT_REPO_ROOT
for v in hs3 hs4 hs7; do
  echo "== $v"
  SF_LIB_PATH=$PWD/starfish_amd/libstarfish_amd_$v.so timeout 300 python tools/bench_potrf.py 4096 128 2 3 2>&1 | grep "^N=\|clock during"
done
