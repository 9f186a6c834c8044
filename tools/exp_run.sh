cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3n
(time python -m pytest tests -m gpu -x -q --durations=8) > gpurun_out/r3n/tests.log 2>&1; tail -16 gpurun_out/r3n/tests.log
python bench.py > gpurun_out/r3n/bench_default.json 2> gpurun_out/r3n/bench_default.err; tail -c 300 gpurun_out/r3n/bench_default.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r3n/bench_default.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['sustained_clock_mhz'])
print([ (r['batch'], round(r['value']), round(r['per_eval_efficiency_vs_full_batch'],3)) for r in d['strong_scaling_proxy']['rows']])
print({k:(round(v['value'],1), round(v['ms_per_step'],1), round(v['roofline']['frac'],3)) for k,v in d['other_configs'].items()})
"
