#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_gpu_recovery.py tests/test_gpu_train.py "tests/test_gpu_parallel.py::test_eight_ranks_take_turns_on_the_device_with_the_persistent_kernel_on" -m gpu -q -s) > $O/new_tests.log 2>&1; tail -5 $O/new_tests.log
NAME=r6b/ab_dataflow TAGS="base exp" ROUNDS=3 CASES="4096 32 4 4;4096 64 4 4;4096 16 4 4;4096 4 4 4;4096 2 4 4;8192 4 2 4;8192 4 2 0;16384 4 2 4;16384 8 1 4;16384 8 1 0" bash tools/run_ab.sh > /dev/null 2>&1
timeout 600 python tools/bench_train.py 4096 60 > $O/bench_train.txt 2>&1; tail -8 $O/bench_train.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.log; tail -c 600 $O/bench_err.log; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6b/bench_line.json").read().strip().splitlines()[-1])
r=d["roofline"]
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if not isinstance(v,(dict,str))})
print(d["value"], d["ms_per_step"]); print(d.get("sampler_step")); print(d.get("train"))
PY
