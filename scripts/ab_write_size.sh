#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the headline's forward kernel per library variant (GPU box; scripts/mkvariant.sh makes the variants).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
cp wfa_amd/lib/libwfahip.so /tmp/libwfahip.orig.so
for v in /tmp/libwfahip.orig.so build/variants/*.so; do
  [ -f $v ] || continue
  cp $v wfa_amd/lib/libwfahip.so
  for C in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/ws_$C; ( cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/ws_$C -- python3 $REPO/bench.py --steps 1 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 > /tmp/ws_$C.log 2>&1 )
    python3 - "$v" $C <<'PY'
import csv, glob, sys
f = sorted(glob.glob(f"/tmp/ws_{sys.argv[2]}/**/*counter_collection.csv", recursive=True))
tot = {}
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "wfa_duo_kernel<false" in r["Kernel_Name"] or "wfa_backtrace_kernel" in r["Kernel_Name"] and r["Grid_Size"] == "1000448":
            k = r["Kernel_Name"][:40]
            tot.setdefault(k, []).append(float(r["Counter_Value"]))
print(sys.argv[1].split("/")[-1], sys.argv[2], {k: round(sum(v) / max(1, len(v)) / 1e6, 2) for k, v in tot.items()}, "GB per dispatch (KB counters; FETCH_SIZE not doubled)")
PY
  done
done
cp /tmp/libwfahip.orig.so wfa_amd/lib/libwfahip.so
