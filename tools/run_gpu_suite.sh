cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3final
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r3final/tests.log 2>&1; tail -3 gpurun_out/r3final/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
