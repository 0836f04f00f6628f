import time, sys
sys.path.insert(0, '.')
import torch
torch.zeros(1, device='cuda:0')
import wfa_amd as w
blob, q_off, q_len, t_off, t_len = w.generate_pairs(3, 200, 1000, 0.05)
qs = [bytes(blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])]) for i in range(200)]
ts = [bytes(blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])]) for i in range(200)]
for mode in (1, 3, 2):
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=True), device=0)
    al.AdaptiveReduction(w.DefaultAdaptiveOption)
    al.set_option("pair_fast", mode)
    al.Align(qs[0], ts[0])
    t1 = time.perf_counter()
    for i in range(200):
        al.Align(qs[i], ts[i])
    print("pair_fast", mode, "us per Align:", (time.perf_counter() - t1) / 200 * 1e6, "launches", al.last_timing().n_launches, flush=True)
    al.close()
