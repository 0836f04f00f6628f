#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_m; mkdir -p $OUT
for o in 0 1; do
WFAHIP_DEBUG_TIMING=1 timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --host-entry 0 --latency 0 --opt narrow_long=$o > $OUT/b$o.json 2> $OUT/b$o.err
python3 -c "
import json; d=json.load(open('$OUT/b$o.json')); c=d['config']; print('narrow_long=$o', 'ms', round(d['ms_per_step'],2), 'main', round(c['main_kernel_ms'],2), 'all', round(c['kernel_ms_per_step'],2), 'retried', c['retried_pairs'], 'launches', c['launches_per_step'])"
grep "handed on" $OUT/b$o.err | tail -2
done
