#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_n; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "duo or short_read or full_size_parity" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
for cfg in "c2 0 100000" "c2 1 100000" "c2 0 1000000" "c2 1 1000000"; do set -- $cfg
  timeout 300 python bench.py --config $1 --steps 300 --warmup 3 --cpu-sample 0 --host-entry 0 --latency 0 --opt duo_short=$2 --pairs $3 > $OUT/bench.json 2> $OUT/bench.err
  python3 -c "
import json; d=json.load(open('$OUT/bench.json')); c=d['config']; print('$1 duo_short=$2 pairs=$3', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'fwd', round(c['main_kernel_ms'],4), 'allk', round(c['kernel_ms_per_step'],4), 'launches', c['launches_per_step'], 'retried', c['retried_pairs'], 'ok', c['status_ok'], d['roofline']['kernel'])" || tail -5 $OUT/bench.err
done
