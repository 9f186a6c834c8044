cd $GRAFT_REPO_ROOT
for seq in 3 0 2 3; do
  echo "== seq $seq"
  timeout 300 python tools/bench_potrf.py 4096 128 3 $seq 2>&1 | grep "^N=\|clock during\|max |L"
done
