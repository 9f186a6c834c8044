# round 4, first GPU call: new tests, default bench line (with the fill leg), fill-stage rocprof passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_a; mkdir -p $O
(time python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "cov_fill or non_positive or poisoned") > $O/tests_new.log 2>&1; tail -5 $O/tests_new.log
(time python -m pytest tests/test_bench_launcher.py -m gpu -x -q) > $O/tests_launcher.log 2>&1; tail -5 $O/tests_launcher.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
bash tools/profile_bench.sh r04_a_fill --fill-only --steps 3 --warmup 1 > $O/profile_fill.log 2>&1; tail -5 $O/profile_fill.log
