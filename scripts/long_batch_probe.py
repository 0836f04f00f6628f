#!/usr/bin/env python3
"""Diagnostic (GPU): a batch of many long global pairs (one workgroup per pair must win over the team kernel)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import wfa_amd as w
n, L = int(sys.argv[1]), int(sys.argv[2])
data = w.generate_pairs(seed=9, n_pairs=n, length=L, error_rate=0.05, n_threads=16)
for opts in ({}, {"team_min_len": 0}, {"team_wgs": 5}):
    al = w.New(); al.AdaptiveReduction(w.DefaultAdaptiveOption)
    for k, v in opts.items(): al.set_option(k, v)
    t0 = time.time(); r = al.align_arrays(*data); dt = time.time() - t0
    t = al.last_timing()
    print(f"n={n} L={L} opts={opts}: wall={dt:.3f}s kernel_ms={t.kernel_ms:.1f} launches={t.n_launches} ok={(r.status == 0).sum()}", flush=True)
    al.close()
