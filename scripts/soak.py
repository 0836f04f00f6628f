#!/usr/bin/env python3
"""GPU box: full-size parity soak.  For several seeds / lengths / error rates: the HIP path on a large batch against the
oracle on all host cores, bit-exact on every field and every CIGAR op; each batch is aligned twice (streamed
backtrace: the hand-over between forward and backtrace waves must hold every time).
Usage: scripts/soak.py [n_pairs]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import wfa_amd as w
from oracle import oracle as O
import test_parity_gpu as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
thr = max(8, (os.cpu_count() or 8) // 2)
bad = 0
for seed, length, err, ad in ((101, 1000, 0.05, (10, 50, 1)), (102, 1000, 0.08, (10, 50, 1)), (103, 600, 0.03, (10, 50, 1)),
                              (104, 2000, 0.05, (10, 50, 1)), (105, 1000, 0.05, (20, 100, 1)), (106, 300, 0.10, None),
                              (107, 150, 0.03, None), (108, 100, 0.06, (10, 50, 1)), (109, 230, 0.04, None),
                              # round 3: shapes that exercise wfa_duo_kernel's widen / narrow / park paths and the learned rows / windows
                              (110, 1000, 0.10, (10, 50, 1)), (111, 1500, 0.04, (10, 50, 1)), (112, 400, 0.06, (10, 50, 1)),
                              (113, 1000, 0.02, (10, 50, 1)), (114, 1000, 0.05, (4, 20, 1)), (115, 800, 0.15, (10, 50, 1)),
                              # wfa_lane_kernel (a lane per pair; with 107-109): full generations, rows near its 30-diagonal limit
                              (116, 150, 0.02, (10, 50, 1)), (117, 60, 0.05, None), (118, 200, 0.05, (10, 50, 1)),
                              # round 4: the sliding-window instances (wfa_blk_kernel<.., LONG>): GPU-filling batches of 5-50 kbp reads
                              (119, 8000, 0.05, (10, 50, 1)), (120, 20000, 0.03, (10, 50, 1)), (121, 50000, 0.05, (10, 50, 1)),
                              (122, 5000, 0.10, (10, 50, 1)),
                              # round 6: semi-global batches on wfa_wide_kernel (negative seed = semi-global): one and four waves per pair, two launches and one
                              (-123, 1000, 0.05, (10, 50, 1)), (-124, 300, 0.08, (10, 50, 1)), (-125, 600, 0.05, (10, 50, 1)), (-126, 450, 0.04, None),
                              (-127, 1800, 0.06, (10, 50, 1)), (-128, 150, 0.05, (10, 50, 1))):
    glob = seed > 0
    seed = abs(seed)
    if os.environ.get("SOAK_SEEDS") and str(seed) not in os.environ["SOAK_SEEDS"].split(","):
        continue  # (SOAK_SEEDS=107,108,...: only those shapes)
    nn = n * 1000 // length if length > 1000 else n
    if not glob:
        nn = nn // 10 if ad is not None else nn // 50  # (eight times the cells of the global pair; without wf-adaptive every row is n + m - 1 wide)
    data = w.generate_pairs(seed=seed, n_pairs=nn, length=length, error_rate=err, n_threads=32)
    t0 = time.perf_counter()
    want = O.align_batch(T._oracle_params(glob, ad), *data, n_threads=thr)
    t1 = time.perf_counter()
    al = T._aligner(glob, ad)
    for rep in range(3):  # (the context learns rows / windows from the first call: the later ones take other passes)
        got = al.align_arrays(*data)
        t = al.last_timing()
        try:
            T.assert_batch_equal(got, want, f"seed={seed} glob={glob} L={length} err={err} ad={ad} rep={rep}")
            print(f"ok   seed={seed} {'global' if glob else 'semi-global'} n={nn} L={length} err={err} ad={ad} rep={rep}: lib {t.total_ms:.1f} ms, kind {t.main_kernel_kind}, "
                  f"retried {t.n_retried_pairs}, oracle {t1 - t0:.1f} s on {thr} threads", flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL", str(e)[:300], flush=True)
    al.close()
print("soak done, failures:", bad)
sys.exit(1 if bad else 0)
