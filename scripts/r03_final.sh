#!/bin/bash
# Round-3 closing run (the one script kept per round): the whole -m gpu suite, one bench line per configuration
# (BASELINE configs + the reference's published grid), rocprofv3 kernel-trace + PMC summaries of the three profiled
# configurations.  Usage on the GPU box (through gpurun): bash scripts/r03_final.sh [tag]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03_final}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -12 $OUT/pytest.log
WFA_TEST_OPTS=arena_poison=1 timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "duo or lane or synthetic or fuzz or other_penalties or mid_window or short_read" > $OUT/pytest_poison.log 2>&1; echo "poisoned arenas: pytest rc $?" | tee -a $OUT/pytest_poison.log; tail -2 $OUT/pytest_poison.log
summ() { python3 -c "
import json; d=json.load(open('$OUT/bench_$1.json')); c=d['config']; r=d['roofline']; cb=d.get('cpu_baseline',{})
print('$1', 'value', round(d['value'],2), 'ms', round(d['ms_per_step'],3), 'fwd', round(c['main_kernel_ms'],3), 'allk', round(c['kernel_ms_per_step'],3), 'launches', c['launches_per_step'], 'retried', c['retried_pairs'],
      'h2h', c.get('host_to_host_ms'), 'packed', c.get('host_to_host_packed_ms'), 'align_us', c.get('single_pair_align_us'), 'kernel', r['kernel'], 'frac', round(r['frac'],4), 'traffic', r.get('traffic'), 'stale', r.get('traffic_stale'),
      'cpu1', cb.get('value'), 'cpuall', cb.get('all_cores',{}).get('value'))" || tail -5 $OUT/bench_$1.err; }
timeout 900 python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err; summ c3
timeout 600 python bench.py --config c2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; summ c2
timeout 900 python bench.py --config c4 --host-entry 0 --latency 0 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; summ c4
timeout 900 python bench.py --config c5s > $OUT/bench_c5s.json 2> $OUT/bench_c5s.err; summ c5s
for c in k10 k20 l5 l10 l20; do timeout 900 python bench.py --config $c --host-entry 0 --latency 0 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; summ $c; done
timeout 900 bash scripts/profile_bench.sh ${TAG}_c3 > $OUT/prof_c3.log 2>&1
timeout 900 bash scripts/profile_bench.sh ${TAG}_c2 --config c2 > $OUT/prof_c2.log 2>&1
timeout 1500 bash scripts/profile_bench.sh ${TAG}_c5s --config c5s > $OUT/prof_c5s.log 2>&1
echo profiles done
timeout 2400 python scripts/soak.py > $OUT/soak.log 2>&1; tail -3 $OUT/soak.log
