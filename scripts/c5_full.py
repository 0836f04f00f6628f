#!/usr/bin/env python3
"""BASELINE configs[4] at its stated count: 1e4 x 100 kbp @10 %, semi-global, wf-adaptive 10/50/1, seed 5, one GPU.

Aligned through wfahip_align_batch_device in calls of 500 pairs (the dataset is generated per call with the generator's
first_index: pair i is the same pair whatever the call size).  Checks on every pair: status OK, the CIGAR consumes both
sequences exactly (M+X+D+H = n, M+X+I = m; +-1 where the reference's own overshoot lengthens it), merged ops; the first
8 pairs -- the bench's c5s sample -- against the oracle on host threads.  Prints one JSON line.
Usage (GPU box): python scripts/c5_full.py [n_pairs=10000] [call=500]
"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wfa_amd as w
from wfa_amd import _lib as L

n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
call = int(sys.argv[2]) if len(sys.argv) > 2 else 500
length, err, seed = 100_000, 0.10, 5
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False), device=0)
assert al.AdaptiveReduction(w.DefaultAdaptiveOption) is None
prm = al._params()
stream = torch.cuda.current_stream(dev).cuda_stream
t_align = 0.0
n_ok = n_over = n_ops_total = 0
scores, first8 = [], None
t_wall0 = time.perf_counter()
for first in range(0, n_total, call):
    n = min(call, n_total - first)
    blob, q_off, q_len, t_off, t_len = w.generate_pairs(seed, n, length, err, first_index=first, n_threads=32)
    sum_len = int(q_len.astype(np.int64).sum() + t_len.astype(np.int64).sum())
    ops_cap = sum_len + 2 * n + 1024
    d_blob = torch.from_numpy(blob).to(dev)
    d = [torch.from_numpy(a.view(np.int64 if a.dtype == np.uint64 else np.int32)).to(dev) for a in (q_off, q_len, t_off, t_len)]
    d_rec = torch.zeros((n, L.REC_WORDS), dtype=torch.int32, device=dev)
    d_ops = torch.zeros(ops_cap, dtype=torch.int64, device=dev)
    needed = C.c_uint64()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    L.check(L.lib().wfahip_align_batch_device(al._ctx, C.byref(prm), d_blob.data_ptr(), blob.size, d[0].data_ptr(), d[1].data_ptr(),
                                              d[2].data_ptr(), d[3].data_ptr(), n, int(max(q_len.max(), t_len.max())),
                                              d_rec.data_ptr(), d_ops.data_ptr(), ops_cap, C.byref(needed), stream), "align")
    torch.cuda.synchronize(dev)
    t_align += time.perf_counter() - t0
    rec = d_rec.cpu().numpy().view(np.uint32)
    ok = rec[:, L.REC_STATUS] == 0
    n_ok += int(ok.sum())
    ops_off = rec[:, L.REC_OPS_OFF_LO].astype(np.int64) | (rec[:, L.REC_OPS_OFF_HI].astype(np.int64) << 32)
    ops_len = rec[:, L.REC_OPS_LEN].astype(np.int64)
    hops = d_ops[:int(needed.value)].cpu().numpy().view(np.uint64)
    for i in range(n):
        o = hops[ops_off[i]:ops_off[i] + ops_len[i]]
        let, cnt = (o >> np.uint64(32)).astype(np.uint8), (o & np.uint64(0xFFFFFFFF)).astype(np.int64)
        qu = int(cnt[np.isin(let, list(b"MXDH"))].sum()); tu = int(cnt[np.isin(let, list(b"MXI"))].sum())
        dq, dt = qu - int(q_len[i]), tu - int(t_len[i])
        assert abs(dq) <= 1 and abs(dt) <= 1, (first + i, dq, dt)
        n_over += (dq != 0) or (dt != 0)
        assert not (let[1:] == let[:-1]).any(), first + i  # merged
    n_ops_total += int(ops_len.sum())
    scores.append(rec[:, L.REC_SCORE].copy())
    if first == 0:
        first8 = (blob, q_off[:8].copy(), q_len[:8].copy(), t_off[:8].copy(), t_len[:8].copy(), rec[:8].copy(),
                  [hops[ops_off[i]:ops_off[i] + ops_len[i]].copy() for i in range(8)])
    tm = al.last_timing()
    print(f"[c5_full] pairs {first + n}/{n_total}: this call {time.perf_counter() - t0:.1f} s, launches {tm.n_launches}, retried {tm.n_retried_pairs}, "
          f"arena {tm.arena_bytes / 2**30:.0f} GiB, start level {tm.ladder_start_level}; so far {(first + n) / t_align:.2f} pairs/s", file=sys.stderr, flush=True)
scores = np.concatenate(scores)
# the oracle on the first 8 pairs (the c5s sample)
from oracle import oracle as O
blob, q_off, q_len, t_off, t_len, rec8, ops8 = first8
t0 = time.perf_counter()
want = O.align_batch(O.make_params(global_alignment=False, adaptive=(10, 50, 1)), blob, q_off, q_len, t_off, t_len, n_threads=8)
t_or = time.perf_counter() - t0
same = all(int(rec8[i, L.REC_SCORE]) == int(want.score[i]) and np.array_equal(ops8[i], want.pair_ops(i)) for i in range(8))
print(json.dumps({"workload": f"{n_total} x {length} bp @{err:.0%}, semi-global, wf-adaptive 10/50/1, seed {seed}", "pairs_ok": n_ok,
                  "align_s": t_align, "pairs_per_s": n_total / t_align, "wall_s": time.perf_counter() - t_wall0,
                  "cigar_ops": n_ops_total, "pairs_with_overshoot": int(n_over), "score_min_median_max": [int(scores.min()), int(np.median(scores)), int(scores.max())],
                  "first_8_pairs_equal_oracle": bool(same), "oracle_8_pairs_s_on_8_threads": t_or}))
