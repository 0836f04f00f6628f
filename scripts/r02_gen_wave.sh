#!/bin/bash
export WFAHIP_DEBUG=1  # (the option knobs below are debug / experiment knobs)
# wave mode of the generic kernel: parity tests that run through it, then timings with and without it
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_gen_wave; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_cli.py -m gpu -x -q -k "not full_size and not config5_full and not soak" --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -10 $OUT/pytest.log
timeout 900 python3 - > $OUT/timing.txt 2>&1 <<'PY'
import sys, time, numpy as np
sys.path.insert(0, ".")
import wfa_amd as w
cases = [("semi-global 20000 x 1 kbp @5%", dict(seed=3, n_pairs=20000, length=1000, error_rate=0.05), False, (10, 50, 1)),
         ("global 4000 x 10 kbp @5%", dict(seed=4, n_pairs=4000, length=10000, error_rate=0.05), True, (10, 50, 1)),
         ("global 500 x 50 kbp @5%", dict(seed=6, n_pairs=500, length=50000, error_rate=0.05), True, (10, 50, 1)),
         ("global 20000 x 1 kbp @5%, generic forced", dict(seed=3, n_pairs=20000, length=1000, error_rate=0.05), True, (10, 50, 1))]
for name, gen, glob, ad in cases:
    data = w.generate_pairs(n_threads=8, **gen)
    ref = None
    for wave in (1, 0):
        al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=glob)); al.AdaptiveReduction(w.AdaptiveReductionOption(*ad))
        al.set_option("team_wave", wave)
        if "forced" in name: al.set_option("packed", 0)
        ms = []
        for rep in range(3):
            r = al.align_arrays(*data); t = al.last_timing()
            key = (int(r.score.sum()), int(r.ops.sum() % (1 << 61)), t.ops_written)
            if ref is None: ref = key
            assert key == ref, (name, wave, key, ref)
            if rep: ms.append(t.kernel_ms)
        print(f"{name}: team_wave={wave} kernel ms {np.round(ms, 2)} launches {t.n_launches} kind {t.main_kernel_kind}", flush=True)
        al.close()
PY
cat $OUT/timing.txt | tail -10
