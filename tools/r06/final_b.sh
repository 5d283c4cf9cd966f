# round 6, end-of-round session B: rocprofv3 summaries of configs 2 / 3 / 4 and 8 x 10^6 particles -> gpurun_out/r06_cfg*, r06_8m
cd $GRAFT_REPO_ROOT
bash tools/profile_r06.sh r06_cfg2 2>&1 | tail -25
bash tools/profile_r06.sh r06_cfg3 --config 3 2>&1 | tail -12
bash tools/profile_r06.sh r06_cfg4 --config 4 2>&1 | tail -12
bash tools/profile_r06.sh r06_8m --side 200 2>&1 | tail -12
