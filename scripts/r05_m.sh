#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_m; mkdir -p $OUT
for o in none chunk_pairs=500000 chunk_pairs=334000 chunk_pairs=250000 chunk_pairs=200000; do
  if [ $o = none ]; then OPT=""; else OPT="--opt $o"; fi
  timeout 600 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 $OPT > $OUT/bench_$o.json 2> $OUT/bench_$o.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_$o.json')); c=d['config']; print('$o: value', round(d['value']/1e6,2), 'M ms', round(d['ms_per_step'],3), 'fwd', round(c['main_kernel_ms'],3), 'allk', round(c['kernel_ms_per_step'],3), 'launches', c['launches_per_step'], 'retried', c['retried_pairs'], 'ok', c['status_ok'])" || tail -3 $OUT/bench_$o.err
done
