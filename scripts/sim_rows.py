#!/usr/bin/env python3
"""Occupancy simulation for the variable-lanes-per-pair forward kernel.

Reads per-score bands of the kept M rows from the CPU oracle (test infrastructure, used here as a measuring
instrument only) for a sample of the configs[2] workload and replays them through a model of one wave whose four
16-lane rows hold either one wide pair (64-diagonal window) or two narrow pairs (32-diagonal windows).
"""
import sys, os, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def traces(n_pairs, length=1000, err=0.05, seed=3, adaptive=(10, 50, 1)):
    import wfa_amd
    from oracle import oracle as O
    blob, qo, ql, to, tl = wfa_amd.generate_pairs(seed=seed, n_pairs=n_pairs, length=length, error_rate=err)
    al = O.Aligner(O.make_params(adaptive=adaptive))
    out = []
    for i in range(n_pairs):
        q = bytes(blob[int(qo[i]):int(qo[i]) + int(ql[i])]); t = bytes(blob[int(to[i]):int(to[i]) + int(tl[i])])
        r = al.align(q, t)
        rows = []
        for s in range(0, r.score + 1, 2):
            w = al.wavefront(0, s)
            if w is None:
                rows.append(None)
                continue
            lo, hi, raw = w
            ks = [lo + j for j, x in enumerate(raw) if x]
            rows.append((min(ks), max(ks)) if ks else None)
        out.append(rows)
    return out


def spans(rows):
    """union span of the kept rows i-3..i after step i (what the window must hold), 0 if none"""
    sp = []
    for i in range(len(rows)):
        los = [rows[j][0] for j in range(max(0, i - 3), i + 1) if rows[j]]
        his = [rows[j][1] for j in range(max(0, i - 3), i + 1) if rows[j]]
        sp.append((max(his) - min(los) + 1) if los else 0)
    return sp


def simulate(sp_all, widen_at=27, narrow_at=22, rows_per_wave=4):
    """one wave; returns wave-steps, parks, refills"""
    queue = list(range(len(sp_all)))[::-1]
    parked = []
    slots = [[None, None] for _ in range(rows_per_wave)]  # each entry: [pair, step] or 'W' marker
    pos = {}
    steps = parks = refills = widen_free = 0
    busy_half_steps = 0
    def take():
        nonlocal refills
        if parked:
            refills += 1
            return parked.pop()
        if queue:
            refills += 1
            return [queue.pop(), 0]
        return None
    while True:
        # refill
        for r in slots:
            for h in (0, 1):
                if r[h] is None:
                    r[h] = take()
        if all(r[0] is None and r[1] is None for r in slots):
            break
        steps += 1
        for r in slots:
            # widen / narrow decisions
            for h in (0, 1):
                e = r[h]
                if e is None or e == 'W':
                    continue
                p, i = e
                sp = sp_all[p]
                need = sp[i] if i < len(sp) else 0
                other = r[1 - h]
                if other == 'W':
                    if need <= narrow_at:
                        r[1 - h] = None  # narrow again: free the half (refilled next step)
                else:
                    if need > widen_at:
                        if other is not None:
                            parked.append(other); parks += 1
                        else:
                            widen_free += 1
                        r[1 - h] = 'W'
            for h in (0, 1):
                e = r[h]
                if e is None or e == 'W':
                    continue
                busy_half_steps += 2 if r[1 - h] == 'W' else 1
                e[1] += 1
                if e[1] >= len(sp_all[e[0]]):
                    if r[1 - h] == 'W':
                        r[1 - h] = None
                    r[h] = None
    return steps, parks, refills, busy_half_steps


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    cache = f"/tmp/sim/traces_{n}.pkl"
    if os.path.exists(cache):
        tr = pickle.load(open(cache, "rb"))
    else:
        tr = traces(n)
        pickle.dump(tr, open(cache, "wb"))
    sp_all = [spans(r) for r in tr]
    tot = sum(len(s) for s in sp_all)
    allsp = np.concatenate([np.array(s) for s in sp_all])
    print("pairs", n, "steps/pair", tot / n, "mean span", allsp.mean(), "frac<=22", (allsp <= 22).mean(), "frac<=27", (allsp <= 27).mean(),
          "frac<=59", (allsp <= 59).mean())
    base = tot / 4
    for wa, na in ((27, 22), (27, 25), (27, 18), (25, 20)):
        st, pk, rf, bh = simulate(sp_all, wa, na)
        print(f"widen>{wa} narrow<={na}: wave-steps {st} vs baseline {base:.0f} -> x{base / st:.3f}; parks/pair {pk / n:.2f} refills/pair {rf / n:.2f} half-slot occupancy {bh / (8 * st):.3f}")
