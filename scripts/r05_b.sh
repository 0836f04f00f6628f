#!/bin/bash
# round 5, second GPU call: stripe mode of the team kernel -- parity first, then the configs[4] sample with and without it
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_b; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -q -x --durations=5 -k "team_kernel or long_pair_semiglobal or config5_sample or generic_kernel_wavefronts" > $OUT/team.log 2>&1; echo "team tests rc $?" | tee -a $OUT/team.log; tail -12 $OUT/team.log
for st in 1 0; do
  timeout 600 python bench.py --config c5s --steps 3 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --opt team_stripe=$st > $OUT/bench_c5s_stripe$st.json 2> $OUT/bench_c5s_stripe$st.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_c5s_stripe$st.json')); c=d['config']; print('c5s stripe=$st: value', round(d['value'],2), 'ms', round(d['ms_per_step'],1), 'ok', c['status_ok'], 'kernel_ms', round(c['main_kernel_ms'],1), 'retried', c['retried_pairs'])" || tail -5 $OUT/bench_c5s_stripe$st.err
done
timeout 900 python -m pytest tests/test_shapes_gpu.py tests/test_parity_gpu.py -m gpu -q -x -k "shape or fuzz_short or lane_kernel" > $OUT/rest.log 2>&1; echo "rest rc $?" | tee -a $OUT/rest.log; tail -5 $OUT/rest.log
