#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e; mkdir -p $O
NAME=r6e/ab_waitslow TAGS="base exp" ROUNDS=3 CASES="4096 32 5 4;4096 64 4 4;4096 48 4 4;4096 16 5 4" bash tools/run_ab.sh > /dev/null 2>&1
NAME=r6e/ab_waitslow_rev TAGS="exp base" ROUNDS=2 CASES="4096 32 5 4;4096 64 4 4" bash tools/run_ab.sh > /dev/null 2>&1
(time timeout 3000 python -m pytest tests -m gpu -x -q) > $O/tests.log 2>&1; tail -4 $O/tests.log
