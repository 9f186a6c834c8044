cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r05_f_cfg3 --config cfg3 > gpurun_out/r05_f_cfg3.log 2>&1; tail -2 gpurun_out/r05_f_cfg3.log | cut -c1-300
