#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_e; mkdir -p $OUT
for c in g_T2_team s_T2_wave s_T3_xbuf s_big; do
  timeout 300 python scripts/teamc_probe.py $c 2>&1 | grep "^\[g_\|^\[s_\|workgroup"
done
timeout 600 python bench.py --config c5s --steps 3 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 > $OUT/bench_c5s.json 2> $OUT/bench_c5s.err
python3 -c "
import json; d=json.load(open('$OUT/bench_c5s.json')); c=d['config']; print('c5s: value', round(d['value'],2), 'ms', round(d['ms_per_step'],1), 'ok', c['status_ok'], 'kernel_ms', round(c['main_kernel_ms'],1), 'retried', c['retried_pairs'], 'arena GiB', round(c['arena_gib'],1))" || tail -5 $OUT/bench_c5s.err
timeout 900 bash scripts/team_stamps.sh 8 > $OUT/stamps_c.txt 2>&1; grep "teamc\|wall\|error" $OUT/stamps_c.txt | tail -9
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -x --durations=5 -k "teamc_kernel or team_kernel or long_pair_semiglobal or config5_sample or learned_start" > $OUT/team.log 2>&1; echo "team tests rc $?" | tee -a $OUT/team.log; tail -12 $OUT/team.log
