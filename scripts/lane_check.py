"""Development check of wfa_lane_kernel (option lane): parity against the oracle over short-read shapes, then timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wfa_amd as w
from oracle import oracle as O

FIELDS = ("status", "score", "tbegin", "tend", "qbegin", "qend", "align_len", "matches", "gaps", "gap_regions", "ops_len")

def mk(ad, pen=(4, 6, 2)):
    al = w.New(w.Penalties(*pen), w.Options(GlobalAlignment=True), device=0)
    if ad is not None:
        al.AdaptiveReduction(w.AdaptiveReductionOption(*ad))
    return al

def cmp(got, want, what):
    bad = 0
    for f in FIELDS:
        a, b = getattr(got, f), getattr(want, f)
        if not np.array_equal(a, b):
            idx = np.nonzero(a != b)[0]
            print(f"  FAIL {what}: {f} differs at {len(idx)} pairs, first {idx[:5]}: {a[idx[:5]]} vs {b[idx[:5]]}")
            bad += 1
    if not bad and not np.array_equal(got.ops, want.ops):
        print(f"  FAIL {what}: ops differ")
        bad += 1
    return bad

fails = 0
shapes = [(150, 0.02, 20000), (150, 0.02, 7), (100, 0.06, 30000), (230, 0.03, 5000), (60, 0.1, 20000), (240, 0.08, 8000), (150, 0.15, 8000), (30, 0.05, 70)]
for ad in ((10, 50, 1), None):
    for L, err, n in shapes:
        data = w.generate_pairs(seed=77 + L + n % 91, n_pairs=n, length=L, error_rate=err, n_threads=8)
        want = O.align_batch(O.make_params(4, 6, 2, global_alignment=True, adaptive=ad), *data, n_threads=8)
        al = mk(ad)
        al.set_option("lane", 2)
        al.set_option("arena_poison", 1)
        for rep in range(2):
            got = al.align_arrays(*data)
            t = al.last_timing()
            f = cmp(got, want, f"L={L} err={err} n={n} ad={ad} rep={rep}")
            fails += f
            print(f"L={L} err={err} n={n} ad={ad} rep={rep}: kind {t.main_kernel_kind} retried {t.n_retried_pairs} {'ok' if not f else 'FAIL'}", flush=True)
        al.close()
# ragged + edge entries
rng = np.random.default_rng(5)
qs, ts = [], []
for i in range(5000):
    L = int(rng.integers(1, 240))
    q = bytes(b"ACGT"[c] for c in rng.integers(0, 4, L))
    t = bytearray(q)
    for _ in range(int(rng.integers(0, 1 + L // 10))):
        pos = int(rng.integers(0, len(t))); kind = int(rng.integers(0, 3))
        if kind == 0: t[pos] = b"ACGT"[int(rng.integers(0, 4))]
        elif kind == 1: t.insert(pos, b"ACGT"[int(rng.integers(0, 4))])
        elif len(t) > 1: del t[pos]
    if i % 7 == 0:
        extra = bytes(b"ACGT"[c] for c in rng.integers(0, 4, int(rng.integers(1, 60))))
        t = (extra + bytes(t)) if i % 2 else (bytes(t) + extra)
    t = bytes(t[:240])
    if i % 17 == 5: q = b""
    elif i % 17 == 9: t = t.lower()
    elif i % 17 == 13: q = (q[:len(q) // 2] + b"N" + q[len(q) // 2:])[:240]
    elif i % 23 == 7: q, t = b"A", b"CA"
    qs.append(q), ts.append(t)
data = w.make_blob(qs, ts)
for ad in ((10, 50, 1), None, (5, 10, 1)):
    for pen in ((4, 6, 2), (2, 3, 1), (6, 9, 3)):
        want = O.align_batch(O.make_params(*pen, global_alignment=True, adaptive=ad), *data, n_threads=8)
        al = mk(ad, pen)
        al.set_option("lane", 2); al.set_option("arena_poison", 1)
        got = al.align_arrays(*data)
        t = al.last_timing()
        f = cmp(got, want, f"ragged ad={ad} pen={pen}")
        fails += f
        print(f"ragged ad={ad} pen={pen}: kind {t.main_kernel_kind} retried {t.n_retried_pairs} {'ok' if not f else 'FAIL'}", flush=True)
        al.close()
print("FAILS", fails)
# timing: c2 shape
for n in (100000, 1000000):
    data = w.generate_pairs(seed=2, n_pairs=n, length=150, error_rate=0.02, n_threads=8)
    for lane in (0, 2):
        al = mk((10, 50, 1))
        al.set_option("lane", lane)
        best = 1e9
        for rep in range(6):
            t0 = time.perf_counter(); got = al.align_arrays(*data); dt = time.perf_counter() - t0
            if rep: best = min(best, dt)
        t = al.last_timing()
        print(f"n={n} lane={lane}: best {best*1e3:.3f} ms wall, kernel_ms {t.kernel_ms:.3f} main {t.main_kernel_ms:.3f} kind {t.main_kernel_kind} retried {t.n_retried_pairs}", flush=True)
        al.close()
