#!/usr/bin/env python3
"""Diagnostic (GPU): timing of the long-read semi-global configuration at growing lengths."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import wfa_amd as w
for L in [int(x) for x in sys.argv[1:]] or [10000, 20000, 40000]:
    data = w.generate_pairs(seed=5, n_pairs=8, length=L, error_rate=0.10, n_threads=8)
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
    t0 = time.time(); r = al.align_arrays(*data); dt = time.time() - t0
    t = al.last_timing()
    print(f"L={L} wall={dt:.3f}s kernel_ms={t.kernel_ms:.1f} launches={t.n_launches} retried={t.n_retried_pairs} cells={t.cells_stored} arena_GiB={t.arena_bytes/2**30:.2f} score0={r.score[0]}", flush=True)
    al.close()
