#!/bin/bash
# tools/pmc_kernel.sh [KERNEL_REGEX] [ab_probe args…] — SQ counters of one kernel (two separate --pmc passes, never combined with a
# trace domain), means per launch; default k_nbr_tile on the bench scene
KRE=${1:-k_nbr_tile}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_kernel; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
  "SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --kernel-include-regex "$KRE" --output-format csv -d $O/pmc$i -o pmc$i -- python $R/tools/ab_probe.py --steps 4 --reps 2 --kernels 0,1,4 "$@" > $O/pmc$i.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/pmc*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{k:28s} launches {len(v):3d} mean {sum(v)/len(v):14.1f}")
PY
