#!/bin/bash
# determinism soak of the team kernel on the c5s sample: N repetitions per option set, every result compared with the first
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_team7; mkdir -p $OUT
timeout ${2:-1500} python3 - "$@" > $OUT/soak.txt 2>&1 <<'PY'
import sys, time, numpy as np
sys.path.insert(0, ".")
import wfa_amd as w
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
data = w.generate_pairs(5, 8, 100000, 0.10, n_threads=8)
for spec in sys.argv[3:] or ["team_strict=0", "team_strict=1"]:
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
    for kv in spec.split(","):
        k, v = kv.split("="); al.set_option(k, int(v))
    ref, bad, ms = None, 0, []
    for rep in range(reps):
        r = al.align_arrays(*data); t = al.last_timing()
        key = (t.cells_stored, t.ops_written, int(r.score.sum()), int(r.ops.sum() % (1 << 61)))
        if ref is None: ref = key
        elif key != ref: bad += 1; print(spec, "rep", rep, "DIFFERS", key, "vs", ref, flush=True)
        if rep: ms.append(t.kernel_ms)
    print(f"{spec}: {reps} repetitions, {bad} differ; kernel ms median {np.median(ms):.1f} min {min(ms):.1f} max {max(ms):.1f}", flush=True)
    al.close()
PY
tail -12 $OUT/soak.txt
