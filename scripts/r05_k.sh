#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_k; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -x --durations=5 -k "scout or paged_arena or config5_sample or learned_start" > $OUT/team.log 2>&1; echo "team tests rc $?" | tee -a $OUT/team.log; tail -6 $OUT/team.log
for c in c5s32 c5s; do
WFAHIP_DEBUG_TIMING=1 timeout 900 python bench.py --config $c --steps 2 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
python3 -c "
import json; d=json.load(open('$OUT/bench_$c.json')); c=d['config']; print('$c: value', round(d['value'],2), 'ms', round(d['ms_per_step'],1), 'ok', c['status_ok'], 'kernel_ms', round(c['main_kernel_ms'],1), 'launches', c['launches_per_step'], 'retried', c['retried_pairs'])" || tail -5 $OUT/bench_$c.err
grep "team kernel:" $OUT/bench_$c.err | tail -4
done
