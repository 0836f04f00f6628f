#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_j; mkdir -p $OUT
for c in c5s32; do
timeout 900 python bench.py --config $c --steps 2 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
python3 -c "
import json; d=json.load(open('$OUT/bench_$c.json')); c=d['config']; print('$c: value', round(d['value'],2), 'ms', round(d['ms_per_step'],1), 'ok', c['status_ok'], 'kernel_ms', round(c['main_kernel_ms'],1), 'retried', c['retried_pairs'])" || tail -5 $OUT/bench_$c.err
done
timeout 3300 python -m pytest tests -m gpu -q -x --durations=12 > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc $?" | tee -a $OUT/gpu_tests.log; tail -18 $OUT/gpu_tests.log
