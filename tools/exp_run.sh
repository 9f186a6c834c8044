cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r03_a_cfg2 --config cfg2 > gpurun_out/r03_a_cfg2.log 2>&1
SF_LIB_PATH=$GRAFT_REPO_ROOT/starfish_amd/libstarfish_amd_tuning.so SF_CHOL_UNFUSED=0 bash tools/profile_bench.sh r03_a_cfg2_narrow --config cfg2 > gpurun_out/r03_a_cfg2_narrow.log 2>&1
bash tools/profile_bench.sh r03_a_cfg3 --config cfg3 --steps 2 > gpurun_out/r03_a_cfg3.log 2>&1
tail -3 gpurun_out/r03_a_cfg2.log gpurun_out/r03_a_cfg2_narrow.log gpurun_out/r03_a_cfg3.log
ls gpurun_out/
