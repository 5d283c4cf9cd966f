cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/r05/dl_probe.py 2>&1 | tail -7
HSA_ENABLE_SDMA=0 timeout 300 python tools/r05/dl_probe.py 2>&1 | tail -7
timeout 600 python bench.py --gpus 2 --transport peer --share-devices --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2 ranks shared', d['ms_per_step'], d['n_gpus'], d['ranks'], d['config']['exchange'])"
