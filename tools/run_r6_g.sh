#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6g; mkdir -p $O
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
for b in 16 1 32; do
  SF_DF_VERBOSE=1 SF_DF_TRACE=1 SF_DF_TRACE_FILE=/tmp/df_$b.txt python tools/bench_potrf.py 4096 $b 1 4 > $O/df_trace_b$b.txt 2>&1
  python tools/df_trace.py /tmp/df_$b.txt 250 >> $O/df_trace_b$b.txt 2>&1
done
unset SF_LIB_PATH
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.log; tail -c 300 $O/bench_err.log; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6g/bench_line.json").read().strip().splitlines()[-1])
r=d["roofline"]
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if not isinstance(v,(dict,str))})
print(d["value"], d["ms_per_step"]); print(d.get("sampler_step")); print(d.get("train"))
PY
