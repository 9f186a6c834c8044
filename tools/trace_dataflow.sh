# per-task trace of one dataflow launch and its utilisation timeline (tools/df_trace.py), tuning build:  BATCHES="32 16" [BIN=250] bash tools/trace_dataflow.sh
cd $GRAFT_REPO_ROOT
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
for b in ${BATCHES:-32 16}; do
  SF_DF_VERBOSE=1 SF_DF_TRACE_FILE=/tmp/df_$b.txt python tools/bench_potrf.py 4096 $b 1 4 2>&1 | grep -E "potrf |waiting" | tail -3 > gpurun_out/df_trace_b$b.txt
  python tools/df_trace.py /tmp/df_$b.txt ${BIN:-250} >> gpurun_out/df_trace_b$b.txt 2>&1
done
