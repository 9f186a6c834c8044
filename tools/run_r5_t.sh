cd $GRAFT_REPO_ROOT
O=gpurun_out/r5t; rm -rf $O; mkdir -p $O
for seq in 0 2; do echo "N=4096 B=128 seq=$seq $(timeout 300 python tools/bench_potrf.py 4096 128 3 $seq 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" | tee -a $O/seq.txt; done
for seq in 0 2; do echo "N=16384 B=32 seq=$seq $(timeout 600 python tools/bench_potrf.py 16384 32 2 $seq 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" | tee -a $O/seq.txt; done
for seq in 0 2; do echo "N=8192 B=64 seq=$seq $(timeout 600 python tools/bench_potrf.py 8192 64 2 $seq 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" | tee -a $O/seq.txt; done
for seq in 0 2; do echo "N=3008 B=1600 seq=$seq $(timeout 600 python tools/bench_potrf.py 3008 1600 2 $seq 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" | tee -a $O/seq.txt; done
for seq in 0 2; do echo "N=4096 B=96 seq=$seq $(timeout 300 python tools/bench_potrf.py 4096 96 3 $seq 2>&1 | grep -E 'potrf [0-9]' | awk '{printf "%s ", $4}')" | tee -a $O/seq.txt; done
