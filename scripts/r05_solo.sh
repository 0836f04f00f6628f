#!/bin/bash
# experiment: rows of how many diagonals are better done by the team (pipelined) than by workgroup 0 alone
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_solo; mkdir -p $OUT
for o in none team_solo_max=1024 team_solo_max=256 team_solo_max=128; do
  if [ $o = none ]; then OPT=""; else OPT="--opt $o"; fi
  for c in c5s32 c5s; do
  timeout 600 python bench.py --config $c --steps 2 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 $OPT > $OUT/bench_${c}_$o.json 2> $OUT/bench_${c}_$o.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_${c}_$o.json')); c=d['config']; print('$c $o: value', round(d['value'],2), 'ms', round(d['ms_per_step'],1), 'ok', c['status_ok'], 'retried', c['retried_pairs'])" || tail -3 $OUT/bench_${c}_$o.err
  done
done
