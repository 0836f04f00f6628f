#!/usr/bin/env python3
"""Condense a scripts/profile_bench.sh output directory into a small text summary (for profiles/)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield f, r
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f, r in rows("stats/**/*kernel_stats.csv"):
    print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
print("== per-dispatch (kernel trace) ==")
for f, r in rows("stats/**/*kernel_trace.csv"):
    if "wfa" in r.get("Kernel_Name", ""):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        print(r["Kernel_Name"][:60], "grid", r.get("Grid_Size_X", r.get("Grid_Size")), "wg", r.get("Workgroup_Size_X", r.get("Workgroup_Size")),
              "vgpr", r.get("VGPR_Count"), "sgpr", r.get("SGPR_Count"), "lds", r.get("LDS_Block_Size"), f"{d:.3f} ms")
print("== PMC (per dispatch, summed over dispatches of the same kernel) ==")
acc = collections.defaultdict(float); cnt = collections.Counter()
for f, r in rows("pmc_*/**/*counter_collection.csv"):
    if "wfa" in r.get("Kernel_Name", ""):
        key = (r["Kernel_Name"][:90], r["Counter_Name"], r.get("Grid_Size", ""))
        acc[key] += float(r["Counter_Value"]); cnt[key] += 1
for k in sorted(acc):
    print(k, "sum", acc[k], "dispatches", cnt[k])

import json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
pm = {}
for k in sorted(acc):
    name, counter, grid = k
    key = counter + "_KB" if counter in ("FETCH_SIZE", "WRITE_SIZE") else counter
    pm.setdefault(name.strip(), {}).setdefault(grid, {})[key] = acc[k] / cnt[k]
with open(os.path.join(out, "pmc.json"), "w") as f:
    json.dump({"note": "rocprofv3 --pmc passes (one counter group per pass), value per dispatch, keyed by kernel and grid size; "
                       "FETCH_SIZE / WRITE_SIZE in KB",
               "kernel_sources_sha": bench.kernel_sources_sha(), "kernels": pm}, f, indent=1)
