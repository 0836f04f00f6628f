#!/usr/bin/env python3
"""SURVEY.md section 8(d) footprint stress: the first pairs of BASELINE configs[4] (100 kbp @10 %, semi-global, seed 5) with
wf-adaptive OFF -- every one of the ~3.5e4 wavefronts keeps all n+m-1 = 2e5 diagonals: ~7e9 cells per pair.

Aligned through wfahip_align_batch_device, `call` pairs per call.  Reported: seconds and pairs/s, launches, the arena the
library took, the ladder level it started on.  Checked on EVERY pair: status OK, the CIGAR consumes both sequences exactly
(+-1 where the reference's own overshoot lengthens it), merged ops, score == the CIGAR's own cost under 4/6/2 (X, gap
opens, gap extensions between the flanking I / H runs, which are free; never below it), every M run over equal bases and every X
over different ones.  Pair 0 against the oracle when the host has the memory for it (3 components x 3.5e4 rows x 2e5
diagonals x 4 bytes = 84 GB) -- else the refusal is printed with the numbers.  One JSON line.
Usage (GPU box): python scripts/c5_adaptive_off.py [n_pairs=8] [call=8] [oracle=1]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wfa_amd as w
from wfa_amd import _lib as L

n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 8
call = int(sys.argv[2]) if len(sys.argv) > 2 else 8
want_oracle = int(sys.argv[3]) if len(sys.argv) > 3 else 1
length, err, seed = 100_000, 0.10, 5
X, O_, E = 4, 6, 2
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False), device=0)  # (no AdaptiveReduction call: off)
prm = al._params()
stream = torch.cuda.current_stream(dev).cuda_stream
out = {"workload": f"{n_total} x {length} bp @{err:.0%}, semi-global 4/6/2, wf-adaptive OFF, seed {seed}", "calls": []}
t_align, n_ok, n_over, scores, keep0 = 0.0, 0, 0, [], None
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
    rc = L.lib().wfahip_align_batch_device(al._ctx, C.byref(prm), d_blob.data_ptr(), blob.size, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                                           d[3].data_ptr(), n, int(max(q_len.max(), t_len.max())), d_rec.data_ptr(), d_ops.data_ptr(), ops_cap,
                                           C.byref(needed), stream)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    t_align += dt
    tm = al.last_timing()
    out["calls"].append({"first": first, "pairs": n, "rc": int(rc), "seconds": dt, "launches": int(tm.n_launches), "retried": int(tm.n_retried_pairs),
                         "arena_gib": tm.arena_bytes / 2**30, "ladder_start_level": int(tm.ladder_start_level), "kernel_kind": int(tm.main_kernel_kind)})
    print(f"[c5_adaptive_off] {out['calls'][-1]}", file=sys.stderr, flush=True)
    if rc != 0:
        out["error"] = L.lib().wfahip_last_error(al._ctx).decode() if hasattr(L.lib(), "wfahip_last_error") else str(rc)
        break
    rec = d_rec.cpu().numpy().view(np.uint32)
    n_ok += int((rec[:, L.REC_STATUS] == 0).sum())
    ops_off = rec[:, L.REC_OPS_OFF_LO].astype(np.int64) | (rec[:, L.REC_OPS_OFF_HI].astype(np.int64) << 32)
    ops_len = rec[:, L.REC_OPS_LEN].astype(np.int64)
    hops = d_ops[:int(needed.value)].cpu().numpy().view(np.uint64)
    for i in range(n):
        if rec[i, L.REC_STATUS] != 0:
            continue
        o = hops[ops_off[i]:ops_off[i] + ops_len[i]]
        let, cnt = (o >> np.uint64(32)).astype(np.uint8), (o & np.uint64(0xFFFFFFFF)).astype(np.int64)
        qu = int(cnt[np.isin(let, list(b"MXDH"))].sum()); tu = int(cnt[np.isin(let, list(b"MXI"))].sum())
        dq, dtt = qu - int(q_len[i]), tu - int(t_len[i])
        assert abs(dq) <= 1 and abs(dtt) <= 1, (first + i, dq, dtt)
        n_over += (dq != 0) or (dtt != 0)
        assert not (let[1:] == let[:-1]).any(), first + i  # merged
        # the CIGAR's own cost between its first and last M run (what lies outside is the free flank of a semi-global alignment)
        ms = np.flatnonzero(~np.isin(let, list(b"IH")))  # (a real gap next to a flank merges with it: then cost < score)
        cost = 0
        if ms.size:
            a, b = int(ms[0]), int(ms[-1])
            for l, c in zip(let[a:b + 1], cnt[a:b + 1]):
                cost += X * int(c) if l == ord("X") else (O_ + E * int(c)) if l in (ord("I"), ord("D")) else 0
        q = blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])]; t = blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])]
        v = h = 0; bad = 0
        for l, c in zip(let, cnt):
            c = int(c)
            if l == ord("M"):
                bad += int((q[v:v + c] != t[h:h + c]).sum()); v += c; h += c
            elif l == ord("X"):
                bad += int((q[v:v + c] == t[h:h + c]).sum()) if (v + c <= len(q) and h + c <= len(t)) else 0; v += c; h += c
            elif l == ord("I"):
                h += c
            else:  # D, H
                v += c
        scores.append((int(rec[i, L.REC_SCORE]), cost, bad))
    if first == 0:
        keep0 = (blob[:int(t_off[0]) + int(t_len[0])].copy() if int(t_off[0]) > int(q_off[0]) else blob.copy(), q_off[:1].copy(), q_len[:1].copy(), t_off[:1].copy(),
                 t_len[:1].copy(), int(rec[0, L.REC_SCORE]), hops[ops_off[0]:ops_off[0] + ops_len[0]].copy())
    del d_blob, d, d_rec, d_ops
    torch.cuda.empty_cache()
out.update({"pairs_ok": n_ok, "align_s": t_align, "pairs_per_s": (n_ok / t_align) if t_align else None, "pairs_with_overshoot": int(n_over),
            "score_equals_cigar_cost": sum(1 for s, c, b in scores if s == c), "score_below_cigar_cost": sum(1 for s, c, b in scores if s < c), "pairs_with_a_wrong_base_under_M_or_X": sum(1 for s, c, b in scores if b),
            "scores": [s for s, c, b in scores]})
print(json.dumps(out), flush=True)
if want_oracle and keep0 is not None:
    avail = 0
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            avail = int(line.split()[1]) * 1024
    need = 3 * 35000 * 200000 * 4 * 1.25
    res = {"oracle_pair0": None, "host_mem_available_gib": avail / 2**30, "oracle_needs_gib": need / 2**30}
    if avail > need:
        from oracle import oracle as O
        blob, q_off, q_len, t_off, t_len, score0, ops0 = keep0
        t0 = time.perf_counter()
        want = O.align_batch(O.make_params(global_alignment=False, adaptive=None), blob, q_off, q_len, t_off, t_len, n_threads=1)
        res.update({"oracle_pair0": bool(int(want.score[0]) == score0 and np.array_equal(ops0, want.pair_ops(0))), "oracle_score": int(want.score[0]),
                    "oracle_seconds_one_thread": time.perf_counter() - t0})
    else:
        res["refused"] = "the oracle keeps every wavefront of the pair: not enough host memory"
    print(json.dumps(res), flush=True)
