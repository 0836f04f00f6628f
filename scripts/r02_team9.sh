#!/bin/bash
# c5s sample: kernel time per option set (shipped library)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_team9; mkdir -p $OUT
timeout 900 python3 - "$@" > $OUT/sweep.txt 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
import wfa_amd as w
data = w.generate_pairs(5, 8, 100000, 0.10, n_threads=8)
ref = None
for spec in sys.argv[1:]:
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
    for kv in spec.split(","):
        k, v = kv.split("="); al.set_option(k, int(v))
    ms = []
    for rep in range(4):
        r = al.align_arrays(*data); t = al.last_timing()
        key = (t.cells_stored, t.ops_written, int(r.score.sum()))
        if ref is None: ref = key
        assert key == ref, (spec, key, ref)
        if rep: ms.append(t.kernel_ms)
    print(f"{spec}: kernel ms {np.round(ms, 1)}  arena GiB {t.arena_bytes / 2**30:.0f}", flush=True)
    al.close()
PY
cat $OUT/sweep.txt | tail -12
