#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_i; mkdir -p $OUT
for c in c5s c5s32; do
timeout 900 python bench.py --config $c --steps 2 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
python3 -c "
import json; d=json.load(open('$OUT/bench_$c.json')); c=d['config']; print('$c: value', round(d['value'],2), 'ms', round(d['ms_per_step'],1), 'ok', c['status_ok'], 'kernel_ms', round(c['main_kernel_ms'],1), 'retried', c['retried_pairs'], 'arena GiB', round(c['arena_gib'],1), 'roofline', d['roofline'])" || tail -5 $OUT/bench_$c.err
done
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -x --durations=5 -k "teamc_kernel or team_kernel or long_pair_semiglobal or config5_sample or learned_start" > $OUT/team.log 2>&1; echo "team tests rc $?" | tee -a $OUT/team.log; tail -4 $OUT/team.log
