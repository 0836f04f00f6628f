#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_l; mkdir -p $OUT
timeout 900 python -m pytest tests/test_entries_gpu.py tests/test_cpp_host.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
python - <<'PY'
import time, wfa_amd as w
for L_, e in ((1000, 0.05), (150, 0.02)):
    blob, qo, ql, to, tl = w.generate_pairs(3, 300, L_, e)
    qs = [bytes(blob[int(qo[i]):int(qo[i]) + int(ql[i])]) for i in range(300)]
    ts = [bytes(blob[int(to[i]):int(to[i]) + int(tl[i])]) for i in range(300)]
    al = w.New(); al.AdaptiveReduction(w.DefaultAdaptiveOption)
    for fast in (1, 2, 0):
        al.set_option("pair_fast", fast)
        for i in range(20): al.Align(qs[i], ts[i])
        t0 = time.perf_counter()
        for i in range(300): al.Align(qs[i], ts[i])
        print(f"L={L_} pair_fast={fast}: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per Align")
PY
