#!/bin/bash
# full GPU suite + c5s / c3 / c2 bench lines after the team kernel work
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_team5; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -14 $OUT/pytest.log
for c in c5s c3; do
timeout 900 python bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err
python - $c <<'PY'
import json, sys
c = sys.argv[1]
d = json.loads(open(f"gpurun_out/r02_team5/bench_{c}.json").read().strip().splitlines()[-1]); print(c, "value", d["value"], "ms", d["ms_per_step"], "stale", d["roofline"].get("traffic_stale"))
PY
done
timeout 900 python bench.py --config c5s --pairs 32 --steps 3 --warmup 1 > $OUT/bench_c5s_32.json 2> $OUT/bench_c5s_32.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02_team5/bench_c5s_32.json").read().strip().splitlines()[-1]); print("c5s_32 value", d["value"], "ms", d["ms_per_step"], d["config"].get("retried_pairs"))
PY
