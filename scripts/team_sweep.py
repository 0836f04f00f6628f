#!/usr/bin/env python3
"""Diagnostic (GPU): team kernel time vs workgroups per team on long semi-global pairs."""
import os; os.environ.setdefault("WFAHIP_DEBUG", "1")  # (these are debug / experiment knobs)
import sys, time, numpy as np
sys.path.insert(0, ".")
import wfa_amd as w
L = int(sys.argv[1]); npairs = int(sys.argv[2])
data = w.generate_pairs(seed=5, n_pairs=npairs, length=L, error_rate=0.10, n_threads=8)
for T in [int(x) for x in sys.argv[3:]]:
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
    if T > 0: al.set_option("team_wgs", T)
    else: al.set_option("team_min_len", 0)
    al.set_option("arena_bytes_per_slot", 24 << 30)  # one launch: no retry ladder in the timing
    t0 = time.time(); r = al.align_arrays(*data); dt = time.time() - t0
    t = al.last_timing()
    print(f"L={L} pairs={npairs} T={T}: wall={dt:.2f}s kernel_ms={t.kernel_ms:.1f} launches={t.n_launches} cells={t.cells_stored}", flush=True)
    al.close()
