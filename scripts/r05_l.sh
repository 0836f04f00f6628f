#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_l; mkdir -p $OUT
for c in g_T2_team s_T2_team s_big; do
  timeout 300 python scripts/teamc_probe.py $c 2>&1 | grep "^\[g_\|^\[s_\|workgroup\|rror" | tail -4
done
for o in team_pipe=1 team_pipe=0; do
timeout 900 python bench.py --config c5s --steps 3 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 --opt $o > $OUT/bench_$o.json 2> $OUT/bench_$o.err
python3 -c "
import json; d=json.load(open('$OUT/bench_$o.json')); c=d['config']; print('c5s $o: value', round(d['value'],2), 'ms', round(d['ms_per_step'],1), 'ok', c['status_ok'], 'kernel_ms', round(c['main_kernel_ms'],1), 'launches', c['launches_per_step'], 'retried', c['retried_pairs'])" || tail -5 $OUT/bench_$o.err
done
timeout 2400 python -m pytest tests/test_parity_gpu.py -m gpu -q -x --durations=5 -k "teamc or team_kernel or config5 or learned_start or semiglobal" > $OUT/team.log 2>&1; echo "team tests rc $?" | tee -a $OUT/team.log; tail -8 $OUT/team.log
