#!/bin/bash
# end-of-round session, part 2: rocprofv3 summaries of the three single-GPU configurations and of the 8M-particle scene
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_r02.sh r02_cfg2 2>&1 | tail -25
bash tools/profile_r02.sh r02_cfg3 --config 3 2>&1 | tail -22
bash tools/profile_r02.sh r02_cfg4 --config 4 2>&1 | tail -22
bash tools/profile_r02.sh r02_8m --side 200 2>&1 | tail -22
