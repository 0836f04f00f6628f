"""Debug aid (GPU box): wfa_wide_kernel against the oracle on a small random semi-global batch.  Usage: wide_debug.py x o e [adaptive: 0|1] [pairs] [maxlen]"""
import os, sys, time
os.environ.setdefault("WFAHIP_DEBUG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wfa_amd as w
from oracle import oracle as O
x, o, e = (int(v) for v in sys.argv[1:4])
ad = (10, 50, 1) if len(sys.argv) > 4 and sys.argv[4] == "1" else None
n = int(sys.argv[5]) if len(sys.argv) > 5 else 50
maxlen = int(sys.argv[6]) if len(sys.argv) > 6 else 200
rng = np.random.default_rng(5)
qs, ts = [], []
for i in range(n):
    L = int(rng.integers(1, maxlen + 1))
    a = rng.integers(0, 4, L)
    b = list(a)
    for _ in range(int(L * 0.08)):
        kind, pos = int(rng.integers(0, 3)), int(rng.integers(0, max(1, len(b))))
        if kind == 0 and b: b[pos] = int(rng.integers(0, 4))
        elif kind == 1: b.insert(pos, int(rng.integers(0, 4)))
        elif b: del b[pos]
    if i % 3 == 0: b = list(rng.integers(0, 4, int(rng.integers(0, 60)))) + b + list(rng.integers(0, 4, int(rng.integers(0, 60))))
    if not b: b = [1]
    qs.append(bytes(b"ACGT"[c] for c in a)); ts.append(bytes(b"ACGT"[c] for c in b[:maxlen]))
data = w.make_blob(qs, ts)
al = w.New(w.Penalties(x, o, e), w.Options(GlobalAlignment=False), device=0)
if ad: al.AdaptiveReduction(w.AdaptiveReductionOption(*ad))
al.set_option("wide_min_pairs", 1)
t0 = time.time()
got = al.align_arrays(*data)
want = O.align_batch(O.make_params(x, o, e, global_alignment=False, adaptive=ad), *data, n_threads=4)
tm = al.last_timing()
bad = [i for i in range(n) if int(got.score[i]) != int(want.score[i]) or not np.array_equal(got.pair_ops(i), want.pair_ops(i))]
print(f"pen {x}/{o}/{e} ad {ad}: kind {tm.main_kernel_kind} retried {tm.n_retried_pairs} mismatching pairs {len(bad)} of {n} {bad[:5]} ({time.time() - t0:.2f} s)", flush=True)
for i in bad[:2]:
    print("  q", qs[i][:60], "t", ts[i][:60], "got", int(got.score[i]), O.ops_to_cigar(got.pair_ops(i))[:80], "want", int(want.score[i]), O.ops_to_cigar(want.pair_ops(i))[:80])
