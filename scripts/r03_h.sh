#!/bin/bash
# Round 3, run H: the whole -m gpu suite with wfa_duo_kernel as the default first pass of large batches, bench lines.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_h; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -14 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python3 -c "
import json; d=json.load(open('$OUT/bench_c3.json')); c=d['config']; print('c3', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'fwd', round(c['main_kernel_ms'],3), 'h2h', c.get('host_to_host_ms'), 'packed', c.get('host_to_host_packed_ms'), 'align_us', c.get('single_pair_align_us'), 'frac', round(d['roofline']['frac'],4), 'kernel', d['roofline']['kernel'], 'cpu', d.get('cpu_baseline',{}).get('value'))"
