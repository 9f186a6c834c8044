cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3z
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r3z/tests.log 2>&1; tail -4 gpurun_out/r3z/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r3z/bench_default.json 2> gpurun_out/r3z/bench_default.err; tail -c 200 gpurun_out/r3z/bench_default.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r3z/bench_default.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['sustained_clock_mhz'], d['roofline']['traffic'])
print([ (r['batch'], round(r['value']), round(r['per_eval_efficiency_vs_full_batch'],3)) for r in d['strong_scaling_proxy']['rows']])
print({k:(round(v['value'],1), round(v['ms_per_step'],1), round(v['roofline']['frac'],3)) for k,v in d['other_configs'].items()})
print(d['cpu_baseline']['value'], d['structured_solver']['value'])
"
