#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_o; mkdir -p $OUT
for o in 1 0; do
timeout 300 python bench.py --steps 20 --warmup 2 --cpu-sample 20000 --cpu-all-cores 0 --host-entry 0 --latency 0 --opt prepack=$o > $OUT/b$o.json 2> $OUT/b$o.err
python3 -c "
import json; d=json.load(open('$OUT/b$o.json')); c=d['config']; print('prepack=$o', 'ms', round(d['ms_per_step'],2), 'main', round(c['main_kernel_ms'],2), 'all', round(c['kernel_ms_per_step'],2), 'retried', c['retried_pairs'], 'ok', c['status_ok'], 'cpu match', d['cpu_baseline']['scores_match_gpu'])"
done
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "synthetic or forward_kernel or ragged or unaligned or known_answers or mid_length or full_size_parity or arena_word or mixed_lengths or hand_over" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
