cd $GRAFT_REPO_ROOT
T=r05_m
O=gpurun_out/$T; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_api.py tests/test_gpu_stages.py -x -q) > $O/tests.log 2>&1; tail -2 $O/tests.log
bash tools/profile_bench.sh ${T}_cfg2 --config cfg2 > $O/cfg2.log 2>&1
bash tools/profile_bench.sh ${T}_fill --fill-only --steps 3 --warmup 1 > $O/fill.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_default_bench_line.json 2> $O/bench_default.err; tail -c 200 gpurun_out/${T}_default_bench_line.json
