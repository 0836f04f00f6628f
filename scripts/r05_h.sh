#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_h; mkdir -p $OUT
hipcc -O3 --offload-arch=gfx950 -o /tmp/exchange_probe scripts/probes/exchange_probe.hip 2>&1 | tail -3
{
timeout 60 /tmp/exchange_probe 1 200000 9 8
timeout 60 /tmp/exchange_probe 32 200000 9 8
for T in 32 16 8; do
 for v in 0 1 2 3 4; do
  timeout 60 /tmp/exchange_probe $T 50000 $v 8 8 1
 done
done
timeout 60 /tmp/exchange_probe 32 50000 0 8 14 1
timeout 60 /tmp/exchange_probe 32 50000 1 8 14 1
timeout 60 /tmp/exchange_probe 32 50000 4 8 14 1
timeout 60 /tmp/exchange_probe 32 50000 0 8 4 1
timeout 60 /tmp/exchange_probe 32 50000 0 8 8 0
timeout 60 /tmp/exchange_probe 32 50000 0 8 8 4
timeout 60 /tmp/exchange_probe 32 50000 1 8 8 0
timeout 60 /tmp/exchange_probe 32 50000 0 1 8 1
timeout 60 /tmp/exchange_probe 32 50000 1 1 8 1
timeout 60 /tmp/exchange_probe 32 50000 2 1 8 1
} 2>&1 | tee $OUT/probe.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --kernel-trace --output-format csv -d $R/$OUT/pmc_ic -- python3 $R/bench.py --config c5s --steps 1 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 > $R/$OUT/pmc_ic.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/r05_h/pmc_ic/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "teamc" in r["Kernel_Name"] or "team_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(f, dict(acc), dict(n))
PY
tail -3 $OUT/pmc_ic.log | cut -c1-600
