#!/bin/bash
# Round 3, run A: new scale tests + baseline lines of the reference's published grid on the round-2 kernels.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_a; mkdir -p $OUT
timeout 1700 python -m pytest tests/test_scale_gpu.py -m gpu -x -q --durations=8 > $OUT/pytest_scale.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_scale.log; tail -15 $OUT/pytest_scale.log
for c in k10 k20 l5 l10 l20; do
  timeout 600 python bench.py --config $c --host-entry 0 --latency 0 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_$c.json')); c=d['config']; print('$c', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'fwd', round(c['main_kernel_ms'],3), 'allk', round(c['kernel_ms_per_step'],3), 'launches', c['launches_per_step'], 'retried', c['retried_pairs'], 'kernel', d['roofline']['kernel'], 'frac', round(d['roofline']['frac'],4), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('all_cores',{}).get('value'))" || tail -5 $OUT/bench_$c.err
done
