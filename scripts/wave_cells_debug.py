#!/usr/bin/env python3
"""Diagnostic (GPU): per-pair stored-cell census of the team kernel with wave mode on / off and of the generic kernel."""
import sys, numpy as np
sys.path.insert(0, ".")
import wfa_amd as w
pen = tuple(int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (5, 20, 3)
a = w.generate_pairs(seed=21, n_pairs=24, length=600, error_rate=0.08)
b = w.generate_pairs(seed=22, n_pairs=4, length=4000, error_rate=0.12)
def one(data, i):
    blob, qo, ql, to, tl = data
    return blob, qo[i:i+1].copy(), ql[i:i+1].copy(), to[i:i+1].copy(), tl[i:i+1].copy()
for name, data in (("a", a), ("b", b)):
    for glob, ad in ((True, (10, 50, 1)), (False, (10, 50, 1)), (True, None), (False, (4, 5, 1))):
        res = []
        for opts in ({"team_min_len": 1, "team_wgs": 3, "team_solo_max": 4096, "team_wave": 1},
                     {"team_min_len": 1, "team_wgs": 3, "team_solo_max": 4096, "team_wave": 0},
                     {"team_min_len": 0}):
            al = w.New(w.Penalties(*pen), w.Options(GlobalAlignment=glob))
            if ad: al.AdaptiveReduction(w.AdaptiveReductionOption(*ad))
            al.set_option("packed", 0)
            for k, v in opts.items(): al.set_option(k, v)
            out = []
            for i in range(len(data[1])):
                r = al.align_arrays(*one(data, i)); out.append((al.last_timing().cells_stored, int(r.score[0])))
            res.append(out)
            al.close()
        for i in range(len(data[1])):
            if len({r[i] for r in res}) > 1:
                print(f"{name} glob={glob} ad={ad} pair {i}: (cells, score) wave/solo/generic {[r[i] for r in res]}", flush=True)
print("done")
