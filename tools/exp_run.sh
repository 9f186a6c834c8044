cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r03_c_cfg2 --config cfg2 > gpurun_out/r03_c_cfg2.log 2>&1
bash tools/profile_bench.sh r03_c_cfg3 --config cfg3 --steps 2 > gpurun_out/r03_c_cfg3.log 2>&1
python bench.py > gpurun_out/r03_c_default_bench_line.json 2> gpurun_out/r03_c_default.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r03_c_default_bench_line.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['sustained_clock_mhz'])
print([ (r['batch'], round(r['value']), round(r['per_eval_efficiency_vs_full_batch'],3)) for r in d['strong_scaling_proxy']['rows']])
print({k:(round(v['value'],1), round(v['ms_per_step'],1), round(v['roofline']['frac'],3)) for k,v in d['other_configs'].items()})
"
for t in r03_c_cfg2 r03_c_cfg3; do python -c "
import json
d=json.load(open('gpurun_out/${t}_pmc_summary.json'))['_k_chol_panel_all']; print('$t', d['hbm_GB_per_step'], d['launches_per_step'])
"; cat gpurun_out/$t/bench_plain.json | python -c "import json,sys; d=json.load(sys.stdin); print('plain', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['sustained_clock_mhz'])"; done
