#!/usr/bin/env python3
"""Print the kernel timeline (ms, relative to the dominant forward kernel's start) of the LAST bench step in a
rocprofv3 --kernel-trace csv.  Usage: scripts/timeline.py <dir-with-*_kernel_trace.csv> [forward-kernel substring]"""
import csv, glob, sys

pat = sys.argv[2] if len(sys.argv) > 2 else "wfa_blk_kernel"
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[max(0, i0 - int(sys.argv[3]) if len(sys.argv) > 3 else i0 - 3):]:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{a:9.3f} {b:9.3f} {b - a:8.3f}  q{r['Queue_Id']}  {r['Kernel_Name'][:70]}")
