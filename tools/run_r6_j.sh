#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6j; mkdir -p $O
timeout 1500 python tools/shared_gpu_probe.py 1 2 8 > $O/shared_gpu.txt 2>&1; tail -30 $O/shared_gpu.txt | cut -c1-330
for c in "4096 64 4" "4096 60 4" "3008 80 4" "8192 32 2" "8192 24 2"; do for seq in "" 0; do echo "== $c | seq '$seq'"; timeout 300 python tools/bench_potrf.py $c $seq 2>&1 | grep -E "potrf [0-9]"; done; done > $O/auto_rule.txt 2>&1; cat $O/auto_rule.txt | cut -c1-60
