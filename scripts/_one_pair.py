import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import wfa_amd as w
blob, qo, ql, to, tl = w.generate_pairs(seed=5, n_pairs=4, length=1000, error_rate=0.05)
q = bytes(blob[int(qo[0]):int(qo[0]) + int(ql[0])]); t = bytes(blob[int(to[0]):int(to[0]) + int(tl[0])])
al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=True), device=0)
al.AdaptiveReduction(w.DefaultAdaptiveOption)
for fast in (1, 2):
    al.set_option("pair_fast", fast)
    for i in range(20): r = al.Align(q, t)
    t0 = time.perf_counter()
    for i in range(200): r = al.Align(q, t)
    dt = (time.perf_counter() - t0) / 200
    tm = al.last_timing()
    print(f"pair_fast={fast}: {dt*1e6:.1f} us per Align, kernel_ms {tm.kernel_ms:.4f}, launches {tm.n_launches}, score {r.Score}")
