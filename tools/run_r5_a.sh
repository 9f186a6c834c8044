# round 5, first GPU call: recovery tests, A/B of the dataflow kernel against round 4's, stress runs (profiles/r05_a_*)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_gpu_recovery.py -x -q) > $O/recovery.log 2>&1; tail -5 $O/recovery.log
LIBS="base tuning" BATCHES="8 16 32 64" bash tools/ab_potrf.sh; cp gpurun_out/ab2.txt $O/ab.txt; cat $O/ab.txt
for nb in "4096 1" "4096 2" "4096 16" "1024 24"; do
  timeout 600 python tools/stress_potrf.py $nb 400 4 2>&1 | tail -1 | tee -a $O/stress.txt
done
