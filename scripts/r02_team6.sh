#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_team6; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "config5 or team or long or semi" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 900 python bench.py --config c5s > $OUT/bench_c5s.json 2> $OUT/bench_c5s.err
timeout 900 python bench.py --config c5s --pairs 32 --steps 3 --warmup 1 > $OUT/bench_c5s_32.json 2> $OUT/bench_c5s_32.err
python - <<'PY'
import json
for c in ("c5s", "c5s_32"):
    d = json.loads(open(f"gpurun_out/r02_team6/bench_{c}.json").read().strip().splitlines()[-1]); print(c, "value", d["value"], "ms", d["ms_per_step"], "retried", d["config"].get("retried_pairs"), "arena_gib", d["config"].get("arena_gib"), "launches", d["config"].get("launches_per_step"))
PY
