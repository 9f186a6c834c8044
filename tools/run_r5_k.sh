cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k; rm -rf $O; mkdir -p $O
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
SF_DIAG_STAMPS=1 timeout 200 python tools/bench_potrf.py 1024 1 1 0 2>&1 | grep k_diag | tee $O/diag_stamps.txt
SF_DIAG_STAMPS=1 timeout 200 python tools/bench_potrf.py 1024 64 1 0 2>&1 | grep k_diag | tee -a $O/diag_stamps.txt
unset SF_LIB_PATH
for i in 1 2 3 4 5; do
SF_BENCH_RANKS_SHARE_GPU=1 OMP_NUM_THREADS=4 timeout 900 python bench.py --gpus 8 --scaling strong --steps 1 --warmup 1 --cpu-sample 0 --no-structured > $O/b8_$i.out 2> $O/b8_$i.err; echo "run $i rc=$?"; python - $O/b8_$i.out <<'P'
import sys,json
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]) if l else {}
print(d.get('error'), d.get('value'))
P
done
