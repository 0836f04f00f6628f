#!/usr/bin/env python3
"""Diagnostic (GPU): team kernel options on the C5 sample (8 x 100 kbp semi-global)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import wfa_amd as w
L = int(sys.argv[1]); 
data = w.generate_pairs(seed=5, n_pairs=8, length=L, error_rate=0.10, n_threads=8)
al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
for kv in sys.argv[2:]:
    for item in kv.split(","):
        k, v = item.split("="); al.set_option(k, int(v))
    t0 = time.time(); r = al.align_arrays(*data); dt = time.time() - t0
    t = al.last_timing()
    print(f"L={L} {kv}: wall={dt:.2f}s kernel_ms={t.kernel_ms:.1f} launches={t.n_launches}", flush=True)
