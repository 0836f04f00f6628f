#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_j; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_scale_gpu.py -m gpu -x -q -k "mid_window or wide_band or grid or pilot or forward_kernel_variants or small_arena or learned" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -12 $OUT/pytest.log
for c in k10 k20 l20; do
  WFAHIP_DEBUG_TIMING=1 timeout 600 python bench.py --config $c --host-entry 0 --latency 0 --steps 20 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_$c.json')); c=d['config']; print('$c', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'fwd', round(c['main_kernel_ms'],3), 'allk', round(c['kernel_ms_per_step'],3), 'launches', c['launches_per_step'], 'retried', c['retried_pairs'], 'kernel', d['roofline']['kernel'], 'frac', round(d['roofline']['frac'],4), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('all_cores',{}).get('value'))" || tail -5 $OUT/bench_$c.err
done
