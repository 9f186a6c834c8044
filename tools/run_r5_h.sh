cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; rm -rf $O; mkdir -p $O
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
SF_WIDE_STAMPS=1 timeout 200 python tools/bench_potrf.py 4096 128 1 2 > $O/stamps_potrf.txt 2>&1
tail -48 $O/stamps_potrf.txt | head -46
SF_WIDE_STAMPS=1 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-structured --no-extra-legs > $O/stamps_bench.txt 2>&1
grep -B2 -A45 "wide launches" $O/stamps_bench.txt | tail -47
