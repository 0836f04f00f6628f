#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-160
t=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wfa" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows[-14:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    print(f"{(s - t0) / 1e6:9.3f} ms  +{(e - s) / 1e6:8.3f} ms  stream {r.get('Stream_Id', r.get('Queue_Id'))}  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size_X', r.get('Grid_Size'))}")
PY
