# round 5: full -m gpu suite, default bench line, rocprofv3 kernel stats + PMC of cfg 2 (profiles/r05_c_*)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; rm -rf $O; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q) > $O/tests.log 2>&1; tail -4 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
bash tools/profile_bench.sh r05_c_cfg2 --config cfg2 > $O/profile.log 2>&1; tail -3 $O/profile.log
