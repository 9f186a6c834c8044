cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; rm -rf $O; mkdir -p $O
SF_BENCH_RANKS_SHARE_GPU=1 OMP_NUM_THREADS=4 timeout 900 python bench.py --gpus 8 --scaling strong --steps 1 --warmup 1 --cpu-sample 0 --no-structured > $O/b8.out 2> $O/b8.err; echo rc=$?; tail -c 1500 $O/b8.out; grep -v "^\[W\|^W0\|^\*\*\*" $O/b8.err | grep -i "error\|Traceback\|File \"/tmp" | head -30
# cfg-5 sized matrices: launch sequences vs dataflow at the batches of a strong split
for b in 4 8 16; do for seq in 0 1 2 4; do
  echo "N=16384 B=$b seq=$seq $(timeout 300 python tools/bench_potrf.py 16384 $b 2 $seq 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" | tee -a $O/cfg5_seq.txt
done; done
for b in 8 16 32; do for seq in 0 2 4; do
  echo "N=8192 B=$b seq=$seq $(timeout 300 python tools/bench_potrf.py 8192 $b 2 $seq 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" | tee -a $O/cfg5_seq.txt
done; done
./tools/probes/write_unaligned > $O/write_unaligned.jsonl; cat $O/write_unaligned.jsonl
export SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so
for g in 32 8 4 2 1; do echo "G=$g $(SF_FILL_BAND_G=$g python tools/bench_fill.py 2>/dev/null | tail -1)" | tee -a $O/fill_g.txt; done
echo "N=3000 $(python tools/bench_fill.py 3000 128 2>/dev/null | tail -1)" | tee -a $O/fill_g.txt
NN=4096 BB=128 SEQ=2 bash tools/trace_potrf.sh > $O/timeline_b128_wide.txt 2>&1
