cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_tl; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --steps 4 --warmup 2 --cpu-sample 0 --host-entry 0 --latency 0 --cpu-all-cores 0 --opt ${LANE_OPT:-lane=2} > $OUT/stats.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
prev = None
for r in rows[-30:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0
    print(f"gap {gap:8.1f} us  dur {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:50]}")
    prev = e
PY
