#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_q; mkdir -p $OUT
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "duo or several_chunks or unaligned_blob or ragged" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
for o in "overlap=0" "overlap=1" "overlap=1 --opt chunk_pairs=250000" "chunk_pairs=500000"; do
  timeout 300 python bench.py --steps 20 --warmup 2 --cpu-sample 0 --host-entry 0 --latency 0 --opt $o > $OUT/bench.json 2> $OUT/bench.err
  python3 -c "
import json; d=json.load(open('$OUT/bench.json')); c=d['config']; print('$o', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'fwd', round(c['main_kernel_ms'],3), 'allk', round(c['kernel_ms_per_step'],3), 'launches', c['launches_per_step'], 'retried', c['retried_pairs'], 'ok', c['status_ok'])" || tail -5 $OUT/bench.err
done
bash scripts/r03_p.sh 2>&1 | tail -9
