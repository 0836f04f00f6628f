#!/bin/bash
# instruction mix of the configs[4] sample's kernel (PMC passes)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_h; mkdir -p $OUT
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$OUT/pmc_$i -- python3 $R/bench.py --config c5s --steps 1 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 > $R/$OUT/pmc_$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r05_h/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "teamc" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print({k: (v / n[k]) for k, v in acc.items()}, dict(n))
PY
