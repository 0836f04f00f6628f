#!/usr/bin/env python3
"""Diff the HIP path against the oracle on a seeded synthetic batch (run on the GPU box)."""
import argparse, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wfa_amd as w
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=200000)
ap.add_argument("--len", type=int, default=1000)
ap.add_argument("--err", type=float, default=0.05)
ap.add_argument("--seed", type=int, default=3)
ap.add_argument("--semi", action="store_true")
ap.add_argument("--no-adaptive", action="store_true")
ap.add_argument("--threads", type=int, default=64)
a = ap.parse_args()
ad = None if a.no_adaptive else (10, 50, 1)
data = w.generate_pairs(a.seed, a.n, a.len, a.err, n_threads=32)
al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=not a.semi), device=0)
if ad: al.AdaptiveReduction(w.AdaptiveReductionOption(*ad))
t0 = time.time(); got = al.align_arrays(*data); t1 = time.time()
tm = al.last_timing()
print(f"gpu: {t1-t0:.3f}s wall, kernel {tm.kernel_ms:.1f} ms, launches {tm.n_launches}, retried {tm.n_retried_pairs}, "
      f"cells {tm.cells_stored}, ops {tm.ops_written}, arena {tm.arena_bytes/2**30:.2f} GiB -> {a.n/tm.kernel_ms*1e3:.0f} pairs/s (kernel)")
t0 = time.time(); want = O.align_batch(O.make_params(global_alignment=not a.semi, adaptive=ad), *data, n_threads=a.threads); t1 = time.time()
print(f"oracle: {t1-t0:.3f}s with {a.threads} threads -> {a.n/(t1-t0):.0f} pairs/s")
bad = np.zeros(a.n, bool)
for f in ("status", "score", "tbegin", "tend", "qbegin", "qend", "align_len", "matches", "gaps", "gap_regions", "ops_len"):
    d = getattr(got, f) != getattr(want, f)
    if d.any(): print("field", f, "differs at", int(d.sum()), "pairs, first", np.nonzero(d)[0][:8])
    bad |= d
if not bad.any() and not np.array_equal(got.ops, want.ops):
    for i in range(a.n):
        if not np.array_equal(got.pair_ops(i), want.pair_ops(i)): bad[i] = True
print("mismatching pairs:", int(bad.sum()))
for i in np.nonzero(bad)[0][:3]:
    print(i, "gpu", got.score[i], O.ops_to_cigar(got.pair_ops(i))[:200])
    print(i, "ora", want.score[i], O.ops_to_cigar(want.pair_ops(i))[:200])
print("oracle cells/pair (Set count) mean:", want.cells.sum(axis=1).mean(), " gpu live cells/pair:", tm.cells_stored / a.n)
