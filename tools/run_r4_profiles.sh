# round-4 evidence for profiles/: kernel stats + PMC of cfg 2 (full batch, and the 16-matrix dataflow case), the dense
# fill, cfg 3 / cfg 5, the default bench line, the batch sweep, per-launch timelines of 16 matrices (launch sequences vs dataflow)
cd $GRAFT_REPO_ROOT
T=${TAG:-r04_k}
bash tools/profile_bench.sh ${T}_cfg2 --config cfg2 > gpurun_out/${T}_cfg2.log 2>&1
bash tools/profile_bench.sh ${T}_cfg2_b16 --config cfg2 --batch 16 > gpurun_out/${T}_cfg2_b16.log 2>&1
bash tools/profile_bench.sh ${T}_fill --fill-only --steps 3 --warmup 1 > gpurun_out/${T}_fill.log 2>&1
bash tools/profile_bench.sh ${T}_cfg3 --config cfg3 > gpurun_out/${T}_cfg3.log 2>&1
bash tools/profile_bench.sh ${T}_cfg5 --config cfg5 > gpurun_out/${T}_cfg5.log 2>&1
python bench.py > gpurun_out/${T}_default_bench_line.json 2> gpurun_out/${T}_default_bench.err
python tools/batch_sweep.py > gpurun_out/${T}_batch_sweep_cfg2.json 2>/dev/null
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
for seq in 0 4; do
  NN=4096 BB=16 SEQ=$seq bash tools/trace_potrf.sh > /dev/null 2>&1
  (grep "potrf " gpurun_out/trace_potrf.log; python tools/trace_summary.py gpurun_out/trace_potrf) > gpurun_out/${T}_timeline_b16_seq$seq.txt 2>&1
done
SF_DF_TRACE=1 SF_DF_VERBOSE=1 python tools/bench_potrf.py 4096 16 1 4 2>&1 | grep -E "^ *[0-9]+ \||per workgroup|matrix 0|St/Sr|potrf " > gpurun_out/${T}_dataflow_chain_b16.txt
tail -2 gpurun_out/${T}_cfg2.log; tail -c 600 gpurun_out/${T}_batch_sweep_cfg2.json
