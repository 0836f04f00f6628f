#!/usr/bin/env python3
"""Diagnostic (GPU): every stored wavefront word of the team kernel (wave mode on / off) against the oracle's dump."""
import sys, numpy as np
sys.path.insert(0, ".")
import wfa_amd as w
from oracle import oracle as O
pen = (5, 20, 3)
b = w.generate_pairs(seed=22, n_pairs=4, length=4000, error_rate=0.12)
blob, qo, ql, to, tl = b
i = 2
q = bytes(blob[qo[i]:qo[i] + ql[i]]); t = bytes(blob[to[i]:to[i] + tl[i]])
oa = O.Aligner(O.make_params(*pen, global_alignment=False, adaptive=(10, 50, 1)))
r = oa.align(q, t); od = oa.dump()
want = {c: {s: {lo + j: v for j, v in enumerate(raw) if v} for s, (lo, hi, raw) in d.items()} for c, d in od.items()}
want = {c: {s: r_ for s, r_ in d.items() if r_} for c, d in want.items()}
print("oracle score", r.score, "cells", sum(len(x) for d in want.values() for x in d.values()))
for wave in (1, 0):
    al = w.New(w.Penalties(*pen), w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.AdaptiveReductionOption(10, 50, 1))
    for k, v in (("packed", 0), ("team_min_len", 1), ("team_wgs", 3), ("team_solo_max", 4096), ("team_wave", wave)):
        al.set_option(k, v)
    wf, res = al.debug_wavefronts(q, t)
    print("wave", wave, "score", res.Score, "cells", sum(len(x) for d in wf.values() for x in d.values()))
    nbad = 0
    for c in "MID":
        for s in sorted(set(wf[c]) | set(want[c])):
            a_, b_ = wf[c].get(s, {}), want[c].get(s, {})
            if a_ != b_:
                ks = sorted(k for k in set(a_) | set(b_) if a_.get(k) != b_.get(k))
                if nbad < 12:
                    print(f"  {c} s={s}: range dev [{min(a_, default=None)},{max(a_, default=None)}] oracle [{min(b_, default=None)},{max(b_, default=None)}] differ at k={ks[:8]} dev {[a_.get(k) for k in ks[:8]]} oracle {[b_.get(k) for k in ks[:8]]}")
                nbad += 1
    print("  differing rows:", nbad)
    al.close()
