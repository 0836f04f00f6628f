#!/usr/bin/env python3
"""GPU box: a batch that needs several chunks (3e6 x 1 kbp: the arenas of one chunk take 35 % of HBM), which is where
the streamed backtrace runs by default -- against the oracle on all host cores, bit-exact, twice."""
import os, sys, time
os.environ["WFAHIP_NO_UPLOAD_OVERLAP"] = "1"  # one device call for the whole batch (the sliced host entry would make four single-chunk calls)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import wfa_amd as w
from oracle import oracle as O
import test_parity_gpu as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
thr = max(8, (os.cpu_count() or 8) // 2)
data = w.generate_pairs(seed=201, n_pairs=n, length=1000, error_rate=0.05, n_threads=32)
t0 = time.perf_counter()
want = O.align_batch(T._oracle_params(True, (10, 50, 1)), *data, n_threads=thr)
print(f"oracle {time.perf_counter() - t0:.1f} s on {thr} threads", flush=True)
al = T._aligner(True, (10, 50, 1))
bad = 0
for rep in range(2):
    got = al.align_arrays(*data)
    t = al.last_timing()
    try:
        T.assert_batch_equal(got, want, f"rep {rep}")
        print(f"ok   rep {rep}: lib {t.total_ms:.1f} ms, {t.n_main_launches} forward launches, retried {t.n_retried_pairs}", flush=True)
    except AssertionError as e:
        bad += 1
        print("FAIL", str(e)[:300], flush=True)
al.close()
sys.exit(1 if bad else 0)
