# two ranks sharing the GPU over the peer transport (a functional run, never a scaling number): the exchange figures with the
# applies beside the all-reduce and without (SALVA_HIP_NO_SPEC_DIST=1), 5 + 40 steps so that the capped regime is in
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r06_s10}; mkdir -p $O
for mode in spec nospec; do
  ( [ $mode = nospec ] && export SALVA_HIP_NO_SPEC_DIST=1; timeout 900 python bench.py --gpus 2 --transport peer --share-devices --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_2ranks_$mode.json 2> $O/bench_2ranks_$mode.err )
  python - <<PY
import json
l=[x for x in open("$O/bench_2ranks_$mode.json") if x.startswith("{")]
d=json.loads(l[-1])
print("$mode", "ms/step", round(d["ms_per_step"],3), "exchange", d["config"]["exchange"], "iters", [i[0] for i in d["iters"]][-8:])
print("  per step ms", d["per_step_ms"][-8:])
PY
done
