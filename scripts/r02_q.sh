#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_q; mkdir -p $OUT
timeout 600 python -m pytest tests/test_entries_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for i in 1 2; do timeout 300 python bench.py --steps 10 --cpu-sample 0 --latency 0 > $OUT/b$i.json 2>$OUT/b$i.err; python3 -c "
import json; d=json.load(open('$OUT/b$i.json')); c=d['config']; print('h2h', c['host_to_host_ms'], 'pack', c['host_pack_ms'], 'packed', c['host_to_host_packed_ms'], c['host_to_host_packed_scores_match'])"; done
