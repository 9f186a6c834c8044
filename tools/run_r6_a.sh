#!/bin/bash
# round 6, first GPU call: the new tests, same-box A/B of the persistent kernel (round-5 build vs this one), B = 64 on
# sequence 4 vs 0, train timing, then the whole GPU suite and the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; mkdir -p $O
(time timeout 1200 python -m pytest tests/test_gpu_recovery.py tests/test_gpu_train.py tests/test_gpu_parallel.py -m gpu -x -q -s) > $O/new_tests.log 2>&1; tail -5 $O/new_tests.log
NAME=r6a/ab_dataflow TAGS="base exp" ROUNDS=3 CASES="4096 16 4 4;4096 32 4 4;4096 64 4 4;4096 64 4 0;4096 8 4 4;4096 1 6 4;2048 16 6 4;16384 4 2 4;16384 4 2 0" bash tools/run_ab.sh > /dev/null 2>&1
tail -n 40 gpurun_out/r6a/ab_dataflow.txt
timeout 600 python tools/bench_train.py 4096 60 > $O/bench_train.txt 2>&1; tail -8 $O/bench_train.txt
(time timeout 3000 python -m pytest tests -m gpu -x -q) > $O/tests.log 2>&1; tail -4 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.log; tail -c 600 $O/bench_err.log; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6a/bench_line.json").read().strip().splitlines()[-1])
r=d["roofline"]
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if not isinstance(v,(dict,str))})
print(d["value"], d["ms_per_step"]); print(d.get("sampler_step")); print(d.get("train"))
PY
