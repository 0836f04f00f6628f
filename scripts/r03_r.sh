#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_r; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "config5 or team or long_pair or learned or small_arena" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
for o in "arena_budget_pct=60" "arena_budget_pct=80"; do
  timeout 600 python bench.py --config c5s --pairs 32 --steps 2 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --opt $o > $OUT/bench.json 2> $OUT/bench.err
  python3 -c "
import json; d=json.load(open('$OUT/bench.json')); c=d['config']; print('c5s x32 $o', 'value', round(d['value'],2), 'ms', round(d['ms_per_step'],1), 'fwd', round(c['main_kernel_ms'],1), 'allk', round(c['kernel_ms_per_step'],1), 'launches', c['launches_per_step'], 'retried', c['retried_pairs'], 'arena GiB', round(c['arena_gib'],1), d['roofline']['kernel'])" || tail -5 $OUT/bench.err
done
