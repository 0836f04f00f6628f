python3 - <<'PY' &
import sys, time
sys.path.insert(0, '.')
import torch
torch.zeros(1, device='cuda:0')
import wfa_amd as w
blob, q_off, q_len, t_off, t_len = w.generate_pairs(3, 200, 1000, 0.05)
qs = [bytes(blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])]) for i in range(200)]
ts = [bytes(blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])]) for i in range(200)]
al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=True), device=0)
al.AdaptiveReduction(w.DefaultAdaptiveOption)
t0 = time.perf_counter()
for r in range(150):
    for i in range(200):
        al.Align(qs[i], ts[i])
print("us per Align over 30000:", (time.perf_counter() - t0) / 30000 * 1e6, flush=True)
PY
sleep 4.5; rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head -6; sleep 1; rocm-smi --showclocks 2>&1 | grep -i "sclk" | head -2; wait
