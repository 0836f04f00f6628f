#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_g; mkdir -p $OUT
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "duo or (forward_kernel_variants and (opts7 or opts8)) or several_chunks" > $OUT/pytest_duo.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_duo.log; tail -3 $OUT/pytest_duo.log
for cfg in "1 1000000" "0 100000" "1 100000" "1 300000"; do set -- $cfg
  timeout 300 python bench.py --steps 20 --warmup 2 --cpu-sample 0 --host-entry 0 --latency 0 --opt duo=$1 --pairs $2 > $OUT/bench.json 2> $OUT/bench.err
  python3 -c "
import json; d=json.load(open('$OUT/bench.json')); c=d['config']; print('duo=$1 pairs=$2', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'fwd', round(c['main_kernel_ms'],3), 'allk', round(c['kernel_ms_per_step'],3), 'launches', c['launches_per_step'], 'retried', c['retried_pairs'], 'ok', c['status_ok'])" || tail -5 $OUT/bench.err
done
