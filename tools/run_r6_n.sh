#!/bin/bash
cd $GRAFT_REPO_ROOT
NAME=r6n_ab_norelease TAGS="exp norel" ROUNDS=3 CASES="4096 16 5 4;4096 32 4 4;4096 64 3 4;4096 1 8 4;2048 16 8 4" bash tools/run_ab.sh > /dev/null 2>&1
grep -E "^==|max .L|potrf" gpurun_out/r6n_ab_norelease.txt | tail -5
