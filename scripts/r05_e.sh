#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_e; mkdir -p $OUT
for c in g_T2_team s_T2_team s_T2_wave s_T3_xbuf s_big; do
  timeout 300 python scripts/teamc_probe.py $c 2>&1 | grep -v "amdgpu.ids\|loop not unrolled\|wfa_reg_kernel\|\^\|warning generated\|In file included\|XCC ids\|teams of" | tail -8
done
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -x --durations=5 -k "teamc_kernel" > $OUT/teamc.log 2>&1; echo "teamc tests rc $?" | tee -a $OUT/teamc.log; tail -15 $OUT/teamc.log
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -q -x --durations=5 -k "team_kernel or long_pair_semiglobal or config5_sample or generic_kernel_wavefronts or learned_start" > $OUT/team.log 2>&1; echo "team tests rc $?" | tee -a $OUT/team.log; tail -12 $OUT/team.log
for c in 1; do
  timeout 600 python bench.py --config c5s --steps 3 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --opt team_compact=$c > $OUT/bench_c5s_compact$c.json 2> $OUT/bench_c5s_compact$c.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_c5s_compact$c.json')); c=d['config']; print('c5s compact=$c: value', round(d['value'],2), 'ms', round(d['ms_per_step'],1), 'ok', c['status_ok'], 'kernel_ms', round(c['main_kernel_ms'],1), 'retried', c['retried_pairs'], 'arena GiB', round(c['arena_gib'],1), 'cells/pair', c['wf_cells_per_pair'])" || tail -5 $OUT/bench_c5s_compact$c.err
done
