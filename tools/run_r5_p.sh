cd $GRAFT_REPO_ROOT
O=gpurun_out/r5p; rm -rf $O; mkdir -p $O
rocm-smi --showmaxpower 2>&1 | grep -i "max" | tee $O/power.txt
( python bench.py --steps 200 --warmup 5 --cpu-sample 0 --no-structured --no-extra-legs > $O/bench.json 2>/dev/null ) &
BP=$!
for i in $(seq 1 70); do echo "$(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks 2>&1 | grep -i "Package Power\|sclk" | sed 's/.*: //' | tr '\n' ' ')"; sleep 0.15; done > $O/samples.txt
wait $BP
sort -k3 -n -t' ' $O/samples.txt | tail -3; awk '{print $NF}' $O/samples.txt | sort -n | tail -5 | tr '\n' ' '; echo
grep -c . $O/samples.txt
python -c "import json; d=json.loads(open('$O/bench.json').readlines()[-1]); print(d['ms_per_step'], d['roofline']['sustained_clock_mhz'])"
