#!/bin/bash
# the sieve launch: 32 x 100 kbp pairs with and without it; parity of a 12-pair batch of 20 kbp pairs through it
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_team10; mkdir -p $OUT
timeout 1500 python3 - > $OUT/sieve.txt 2>&1 <<'PY'
import sys, time, numpy as np
sys.path.insert(0, ".")
import wfa_amd as w
data = w.generate_pairs(5, 32, 100000, 0.10, n_threads=8)
ref = None
for sieve in (1, 0):
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
    al.set_option("sieve", sieve); al.set_option("arena_poison", 0)
    for rep in range(3):
        t0 = time.time(); r = al.align_arrays(*data); t = al.last_timing(); dt = time.time() - t0
        key = (t.cells_stored, t.ops_written, int(r.score.sum()), int(r.ops.sum() % (1 << 61)))
        if ref is None: ref = key
        print(f"sieve={sieve} rep {rep}: wall {dt:.3f} s kernel_ms {t.kernel_ms:.0f} first launch {t.main_kernel_ms:.0f} launches {t.n_launches} retried {t.n_retried_pairs} {'same' if key == ref else 'DIFFERS'}", flush=True)
    al.close()
PY
cat $OUT/sieve.txt | tail -8
