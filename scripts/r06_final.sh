#!/bin/bash
# Round-6 closing run (the one script kept per round): the whole -m gpu suite, one bench line per configuration
# (BASELINE configs + the reference's published grid + the GPU-filling variants c2m / L5), rocprofv3 kernel-trace + PMC
# summaries of the profiled configurations, the team-kernel soak.  Usage on the GPU box (through gpurun):
#   bash scripts/r06_final.sh [tag] [what...]     what: tests lines profiles stress soak (default: these) c5full smoke
# Order at the end of a round: profiles first, scripts/collect_profiles.sh + commit, THEN lines -- bench.py names the newest committed PMC profile of the
# same kernel sources (roofline.traffic_stale says when it could not)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06_final}; shift || true
WHAT=${*:-tests lines profiles stress soak}
OUT=gpurun_out/$TAG; mkdir -p $OUT
has() { case " $WHAT " in *" $1 "*) return 0;; esac; return 1; }
if has tests; then
  timeout 3000 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -12 $OUT/pytest.log
  WFA_TEST_OPTS=arena_poison=1 timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_shapes_gpu.py -m gpu -q -k "duo or lane or synthetic or fuzz or other_penalties or mid_window or short_read or long_window or shape" > $OUT/pytest_poison.log 2>&1; echo "poisoned arenas: pytest rc $?" | tee -a $OUT/pytest_poison.log; tail -2 $OUT/pytest_poison.log
fi
summ() { python3 -c "
import json; d=json.load(open('$OUT/bench_$1.json')); c=d['config']; r=d['roofline']; cb=d.get('cpu_baseline',{})
print('$1', 'value', round(d['value'],2), 'ms', round(d['ms_per_step'],3), 'fwd', round(c['main_kernel_ms'],3), 'allk', round(c['kernel_ms_per_step'],3), 'launches', c['launches_per_step'], 'retried', c['retried_pairs'],
      'h2h', c.get('host_to_host_ms'), 'packed', c.get('host_to_host_packed_ms'), 'align_us', c.get('single_pair_align_us'), 'kernel', r['kernel'], 'frac', round(r['frac'],4), 'traffic', r.get('traffic'), 'stale', r.get('traffic_stale'),
      'cpu1', cb.get('value'), 'cpuall', cb.get('all_cores',{}).get('value'), 'others', {k: (round(v.get('value', 0), 1), v.get('status_ok')) for k, v in c.get('other_configs', {}).items()})" || tail -5 $OUT/bench_$1.err; }
if has lines; then
  timeout 900 python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err; summ c3
  for c in c2 c2m p242; do timeout 600 python bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err; summ $c; done
  timeout 1200 python bench.py --config g3 --host-entry 0 --latency 0 > $OUT/bench_g3.json 2> $OUT/bench_g3.err; summ g3
  timeout 900 python bench.py --config c4 --host-entry 0 --latency 0 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; summ c4
  timeout 900 python bench.py --config c5s > $OUT/bench_c5s.json 2> $OUT/bench_c5s.err; summ c5s
  timeout 900 python bench.py --config c5s --pairs 32 --steps 1 --cpu-sample 0 --host-entry 0 --latency 0 > $OUT/bench_c5s32.json 2> $OUT/bench_c5s32.err; cp $OUT/bench_c5s32.json $OUT/bench_x.json; python3 -c "
import json; d=json.load(open('$OUT/bench_c5s32.json')); print('c5s x 32 pairs: value', round(d['value'],2), 'ms', round(d['ms_per_step'],1), 'ok', d['config']['status_ok'])"
  for c in k10 k20 l5 L5 l10 l20; do timeout 900 python bench.py --config $c --host-entry 0 --latency 0 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; summ $c; done
  timeout 300 python scripts/align_latency.py 2>&1 | grep pair_fast | tee $OUT/align_latency.txt
  timeout 600 python bench.py --gpus 1 --force-collective --steps 20 --warmup 3 --host-entry 0 --latency 0 --cpu-sample 0 > $OUT/bench_c3_nccl1.json 2> $OUT/bench_c3_nccl1.err; python3 -c "
import json; d=json.loads([l for l in open('$OUT/bench_c3_nccl1.json') if l.startswith('{')][0]); c=d['config']; print('c3 with the RCCL process group (1 rank): value', round(d['value'],1), 'backend', c['backend'], 'gather ms', c['gather_ms_standalone'], 'complete', c['gathered_records_complete'])"
fi
if has profiles; then
  timeout 900 bash scripts/profile_bench.sh ${TAG}_c3 > $OUT/prof_c3.log 2>&1
  timeout 900 bash scripts/profile_bench.sh ${TAG}_p242 --config p242 > $OUT/prof_p242.log 2>&1
  timeout 900 bash scripts/profile_bench.sh ${TAG}_g3 --config g3 > $OUT/prof_g3.log 2>&1
  timeout 900 bash scripts/profile_bench.sh ${TAG}_c2 --config c2 > $OUT/prof_c2.log 2>&1
  timeout 900 bash scripts/profile_bench.sh ${TAG}_c2m --config c2m > $OUT/prof_c2m.log 2>&1
  timeout 900 bash scripts/profile_bench.sh ${TAG}_l5 --config l5 > $OUT/prof_l5.log 2>&1
  timeout 1200 bash scripts/profile_bench.sh ${TAG}_L5 --config L5 > $OUT/prof_L5.log 2>&1
  timeout 900 bash scripts/profile_bench.sh ${TAG}_l20 --config l20 > $OUT/prof_l20.log 2>&1
  timeout 900 bash scripts/profile_bench.sh ${TAG}_k10 --config k10 > $OUT/prof_k10.log 2>&1
  timeout 1500 bash scripts/profile_bench.sh ${TAG}_c5s --config c5s > $OUT/prof_c5s.log 2>&1
  echo profiles done
fi
if has stress; then
  timeout 1500 python scripts/c5_adaptive_off.py 8 8 1 > $OUT/c5_adaptive_off.json 2> $OUT/c5_adaptive_off.log; tail -2 $OUT/c5_adaptive_off.json | cut -c1-400
fi
if has c5full; then  # configs[4] at its stated count (oracle on the first 8 pairs, properties on all)
  timeout 1500 python scripts/c5_full.py 10000 500 > $OUT/c5_full.json 2> $OUT/c5_full.log; tail -2 $OUT/c5_full.log; cat $OUT/c5_full.json
fi
if has smoke; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
fi
if has soak; then
  timeout 900 python scripts/team_soak.py 120 team_xcd=2 2>&1 | grep -v amdgpu.ids | tee $OUT/team_soak.txt
  timeout 2400 python scripts/soak.py > $OUT/soak.log 2>&1; tail -3 $OUT/soak.log
fi
