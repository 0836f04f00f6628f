#!/usr/bin/env python3
"""Diagnostic (GPU): stored-cell census of long semi-global pairs, HIP path vs the oracle's Set count."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import wfa_amd as w
from oracle import oracle as O
for L in [int(x) for x in sys.argv[1:]] or [20000, 30000, 40000]:
    data = w.generate_pairs(seed=5, n_pairs=2, length=L, error_rate=0.10, n_threads=2)
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
    t0 = time.time(); r = al.align_arrays(*data); dt = time.time() - t0
    t = al.last_timing()
    t1 = time.time(); ref = O.align_batch(O.make_params(global_alignment=False, adaptive=(10, 50, 1)), *data, n_threads=2); do = time.time() - t1
    print(f"L={L} gpu_wall={dt:.3f}s kernel_ms={t.kernel_ms:.1f} launches={t.n_launches} gpu_cells={t.cells_stored} oracle_sets={int(ref.cells.sum())} oracle_s={do:.2f} same_score={np.array_equal(r.score, ref.score)}", flush=True)
    al.close()
