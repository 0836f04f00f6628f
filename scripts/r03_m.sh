#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_m; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_entries_gpu.py tests/test_parity_gpu.py -m gpu -x -q -k "entries or host_entry or entry" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
WFAHIP_DEBUG_TIMING=1 timeout 600 python bench.py --steps 20 --cpu-sample 0 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python3 -c "
import json; d=json.load(open('$OUT/bench_c3.json')); c=d['config']; print('c3', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'fwd', round(c['main_kernel_ms'],3), 'h2h', c.get('host_to_host_ms'), 'packed', c.get('host_to_host_packed_ms'), 'pack', c.get('host_pack_ms'), 'align_us', c.get('single_pair_align_us'))"
grep "host entry\|slice" $OUT/bench_c3.err | tail -24
