#!/usr/bin/env python3
"""Extract the two M-component tables printed in the reference's README (README.md:101-114 global KA1,
README.md:128-140 semi-global KA2) into tests/golden/plot_tables.json: rows of stripped cells.  These are OUTPUTS
of the reference's Plot (wfa_component_plot.go:41-209) rendered as markdown; run here, where /root/reference exists."""
import json, sys
src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/README.md"
lines = open(src, encoding="utf-8").read().split("\n")
def table(start_marker):
    i = next(k for k, l in enumerate(lines) if l.strip() == start_marker)
    rows = []
    for l in lines[i + 1:]:
        if l.startswith("|"):
            cells = [c.strip() for c in l.strip().strip("|").split("|")]
            if all(set(c) <= set(":-") for c in cells):
                continue  # the markdown separator row
            rows.append(cells)
        elif rows:
            break
    return rows
out = {"source": "shenwei356/wfa README.md (v0.4.0 checkout): M-component tables of the global and semi-global examples",
       "ka1_global": table("Global alignment"), "ka2_semiglobal": table("Semi-global alignment")}
json.dump(out, open("tests/golden/plot_tables.json", "w"), ensure_ascii=False, indent=1)
print({k: len(v) for k, v in out.items() if isinstance(v, list)})
