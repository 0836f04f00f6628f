#!/usr/bin/env python3
"""Diagnostic (GPU): PCIe-inclusive rate of the host entry wfahip_align_batch (H2D, kernels, D2H, re-pack)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import wfa_amd as w
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
data = w.generate_pairs(3, n, 1000, 0.05, n_threads=32)
al = w.New(); al.AdaptiveReduction(w.DefaultAdaptiveOption)
for it in range(3):
    t0 = time.time(); r = al.align_arrays(*data); dt = time.time() - t0
    t = al.last_timing()
    print(f"n={n} wall={dt:.3f}s -> {n/dt:.3e} pairs/s  (device part {t.total_ms:.1f} ms)", flush=True)
