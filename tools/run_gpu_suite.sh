cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-suite}; mkdir -p $O
(time timeout 3000 python -m pytest tests -m gpu -x -q) > $O/tests.log 2>&1; tail -4 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
