cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; rm -rf $O; mkdir -p $O
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
for i in 1 2 3 4 5 6; do
env SF_DF_TIMEOUT_S=60 SF_DF_CHECK=1 SF_BENCH_RANKS_SHARE_GPU=1 OMP_NUM_THREADS=4 timeout 900 python bench.py --gpus 8 --scaling strong --single-scaling --steps 1 --warmup 1 --cpu-sample 0 --no-structured > $O/b8_$i.out 2> $O/b8_$i.err; echo "run $i rc=$? t=$SECONDS aborted=$(grep -c 'dataflow ABORTED' $O/b8_$i.err)"
python - $O/b8_$i.out <<'P'
import sys,json
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]) if l else {}
print(d.get('error'), d.get('value'), d.get('ms_per_step'))
P
done
