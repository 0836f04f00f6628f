#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_t; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -10 $OUT/pytest.log
timeout 600 python bench.py --steps 30 --cpu-sample 0 --host-entry 0 --latency 0 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python3 -c "
import json; d=json.load(open('$OUT/bench_c3.json')); c=d['config']; print('c3', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'fwd', round(c['main_kernel_ms'],3), 'allk', round(c['kernel_ms_per_step'],3), 'retried', c['retried_pairs'])"
bash scripts/profile_bench.sh r03_t_c3 > $OUT/prof.log 2>&1; grep -E "duo_kernel<false>|prepack|backtrace" gpurun_out/prof_r03_t_c3/summary.txt | grep -E "AverageNs|WRITE_SIZE|FETCH_SIZE|SQ_INSTS_VALU'" | cut -c1-230
