#!/usr/bin/env python3
"""Diagnostic (GPU): the team kernel (several workgroups per pair) against the oracle on small and long pairs."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import wfa_amd as w
from oracle import oracle as O
FIELDS = ("score", "tbegin", "tend", "qbegin", "qend", "align_len", "matches", "gaps", "gap_regions", "ops_len")
def check(name, data, glob, ad, opts):
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=glob))
    if ad: al.AdaptiveReduction(w.AdaptiveReductionOption(*ad))
    for k, v in opts.items(): al.set_option(k, v)
    t0 = time.time(); got = al.align_arrays(*data); dt = time.time() - t0
    t = al.last_timing()
    want = O.align_batch(O.make_params(global_alignment=glob, adaptive=ad), *data, n_threads=8)
    ok = np.array_equal(got.status, want.status) and all(np.array_equal(getattr(got, f), getattr(want, f)) for f in FIELDS) and np.array_equal(got.ops, want.ops)
    print(f"{name}: {'OK' if ok else 'MISMATCH'} wall={dt:.3f}s kernel_ms={t.kernel_ms:.1f} launches={t.n_launches} cells={t.cells_stored}", flush=True)
    al.close()
    return ok
which = sys.argv[1] if len(sys.argv) > 1 else "small"
if which == "small":
    d = w.generate_pairs(seed=11, n_pairs=40, length=700, error_rate=0.1)
    for glob in (True, False):
        for ad in ((10, 50, 1), None):
            for T in (2, 5):
                for solo in (0, 16, 4096):
                    check(f"700bp glob={glob} ad={ad} T={T} solo_max={solo}", d, glob, ad,
                          {"packed": 0, "team_min_len": 1, "team_wgs": T, "team_solo_max": solo})
    d = w.generate_pairs(seed=12, n_pairs=6, length=5000, error_rate=0.1)
    check("5kbp semi T=3 solo_max=64", d, False, (10, 50, 1), {"team_min_len": 1, "team_wgs": 3, "team_solo_max": 64})
else:
    for L in [int(x) for x in sys.argv[2:]]:
        d = w.generate_pairs(seed=5, n_pairs=2, length=L, error_rate=0.10, n_threads=2)
        check(f"{L}bp semi team", d, False, (10, 50, 1), {})
