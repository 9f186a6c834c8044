# round 5, final evidence on the committed code: -m gpu suite, smoke, default bench line, rocprofv3 kernel stats + PMC of cfg 2 (full batch and
# the 16-matrix dataflow case), cfg 3, cfg 5, the dense fill, batch sweep, dataflow chain trace
cd $GRAFT_REPO_ROOT
T=${TAG:-r05_k}
O=gpurun_out/$T; mkdir -p $O
(time timeout 3000 python -m pytest tests -m gpu -x -q) > $O/tests.log 2>&1; tail -3 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_default_bench_line.json 2> $O/bench_default.err; tail -c 300 gpurun_out/${T}_default_bench_line.json
bash tools/profile_bench.sh ${T}_cfg2 --config cfg2 > $O/cfg2.log 2>&1
bash tools/profile_bench.sh ${T}_cfg2_b16 --config cfg2 --batch 16 > $O/cfg2_b16.log 2>&1
bash tools/profile_bench.sh ${T}_fill --fill-only --steps 3 --warmup 1 > $O/fill.log 2>&1
bash tools/profile_bench.sh ${T}_cfg3 --config cfg3 > $O/cfg3.log 2>&1
bash tools/profile_bench.sh ${T}_cfg5 --config cfg5 > $O/cfg5.log 2>&1
python tools/batch_sweep.py > gpurun_out/${T}_batch_sweep_cfg2.json 2>/dev/null
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
SF_DF_TRACE=1 SF_DF_VERBOSE=1 python tools/bench_potrf.py 4096 16 1 4 2>&1 | grep -E "^ *[0-9]+ \||per workgroup|matrix 0|St/Sr|potrf " > gpurun_out/${T}_dataflow_chain_b16.txt
for nb in "4096 1" "4096 2" "4096 16" "1024 24"; do timeout 600 python tools/stress_potrf.py $nb 400 4 2>&1 | tail -1; done > gpurun_out/${T}_dataflow_stress.txt
tail -2 $O/cfg2.log | cut -c1-200
