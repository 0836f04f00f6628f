#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_g; mkdir -p $OUT
timeout 900 bash scripts/team_stamps.sh 8 > $OUT/stamps.txt 2>&1; grep "teamc [0-9]\] steps\|pipelined steps\|^wall" $OUT/stamps.txt | tail -17 | grep -A1 "stripe(team) [0-9][0-9][0-9]\|^wall" | cut -c1-460
for c in c5s c5s32; do
timeout 900 python bench.py --config $c --steps 3 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
python3 -c "
import json; d=json.load(open('$OUT/bench_$c.json')); c=d['config']; print('$c: value', round(d['value'],2), 'ms', round(d['ms_per_step'],1), 'ok', c['status_ok'])"
done
timeout 2400 python -m pytest tests/test_parity_gpu.py -m gpu -q -x --durations=5 -k "teamc or team_kernel or config5 or learned_start or semiglobal" > $OUT/team.log 2>&1; echo "team tests rc $?" | tee -a $OUT/team.log; tail -4 $OUT/team.log
