import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import wfa_amd as w
from oracle import oracle as O
import test_parity_gpu as T
pen, glob, ad = (3, 3, 0), False, (4, 5, 1)
data = w.generate_pairs(seed=sum(pen), n_pairs=600, length=200, error_rate=0.08)
blob, q_off, q_len, t_off, t_len = data
al = T._aligner(glob, ad, pen)
got = al.align_arrays(*data)
want = O.align_batch(T._oracle_params(glob, ad, pen), *data, n_threads=8)
bad = np.nonzero(got.score != want.score)[0]
print("bad pairs", bad, got.score[bad], want.score[bad])
i = int(bad[0])
q = bytes(blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])]); t = bytes(blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])])
print(q.decode()); print(t.decode())
wf, res = al.debug_wavefronts(q, t)
print("debug path score", res.Score, res.CIGAR(False)[:80])
oa = O.Aligner(O.make_params(*pen, global_alignment=glob, adaptive=ad))
r = oa.align(q, t)
print("oracle score", r.score, r.cigar[:80])
dump = oa.dump()
n_diff = 0
for comp in "MID":
    od = {s: {lo + j: v for j, v in enumerate(raw) if v} for s, (lo, hi, raw) in dump[comp].items()}
    od = {s: r_ for s, r_ in od.items() if r_}
    for s in sorted(set(od) | set(wf[comp])):
        a, b = wf[comp].get(s, {}), od.get(s, {})
        if a != b:
            ks = sorted(k for k in set(a) | set(b) if a.get(k) != b.get(k))
            print(comp, "score", s, "first diffs", [(k, a.get(k), b.get(k)) for k in ks[:6]], "n", len(ks), "gpu range", (min(a) if a else None, max(a) if a else None), "oracle range", (min(b) if b else None, max(b) if b else None))
            n_diff += 1
            if n_diff > 8: sys.exit(0)
