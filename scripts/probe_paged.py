import sys; sys.path.insert(0, ".")
import torch; torch.zeros(1, device="cuda:0")
import wfa_amd as w
data = w.generate_pairs(seed=77, n_pairs=10, length=20000, error_rate=0.10)
for gib in (0, 4, 5, 6, 8):
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
    if gib: al.set_option("mem_limit", gib << 30)
    got = al.align_arrays(*data); t = al.last_timing()
    print(gib, "GiB:", list(got.status), "launches", t.n_launches, "retried", t.n_retried_pairs, "arena MiB", t.arena_bytes >> 20, "cells", t.cells_stored, flush=True)
    al.close()
