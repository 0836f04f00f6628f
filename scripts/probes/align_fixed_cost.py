import time, sys
sys.path.insert(0, '.')
import torch
torch.zeros(1, device='cuda:0')
import wfa_amd as w
blob, q_off, q_len, t_off, t_len = w.generate_pairs(3, 200, 1000, 0.05)
qs = [bytes(blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])]) for i in range(200)]
ts = [bytes(blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])]) for i in range(200)]
al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=True), device=0)
al.AdaptiveReduction(w.DefaultAdaptiveOption)
r = al.Align(qs[0], ts[0])
sc = []
for i in range(200):
    r = al.Align(qs[i], ts[i]); sc.append(r.Score)
print("mean score", sum(sc) / len(sc), "max", max(sc))
t1 = time.perf_counter()
for i in range(200):
    al.Align(qs[i], qs[i])
print("identical us per Align:", (time.perf_counter() - t1) / 200 * 1e6, flush=True)
short = [q[:100] for q in qs]
t1 = time.perf_counter()
for i in range(200):
    al.Align(short[i], short[i])
print("identical 100bp us per Align:", (time.perf_counter() - t1) / 200 * 1e6, flush=True)
al.close()
