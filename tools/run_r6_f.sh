#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_recovery.py -m gpu -q -x) > $O/tests.log 2>&1; tail -3 $O/tests.log
NAME=r6f/ab_codesize TAGS="base exp" ROUNDS=3 CASES="4096 16 5 4;4096 32 5 4;4096 64 4 4;4096 8 5 4;4096 1 8 4;4096 4 6 4;2048 16 8 4;3008 32 5 4;8192 8 2 4" bash tools/run_ab.sh > /dev/null 2>&1
NAME=r6f/ab_codesize_rev TAGS="exp base" ROUNDS=2 CASES="4096 16 5 4;4096 32 5 4" bash tools/run_ab.sh > /dev/null 2>&1
