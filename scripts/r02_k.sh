#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_k; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -10 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err; python3 -c "
import json; d=json.load(open('$OUT/bench_c3.json')); c=d['config']; print('value', d['value'], 'ms', d['ms_per_step'], 'fwd', c['main_kernel_ms'], 'h2h', c['host_to_host_ms'], 'packed', c.get('host_to_host_packed_ms'), 'align_us', c['single_pair_align_us'], 'frac', d['roofline']['frac'])"
