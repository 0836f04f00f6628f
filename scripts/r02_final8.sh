#!/bin/bash
# Round-2 closing run (8): final sources -- the -m gpu suite, streamed-path tests, bench lines, rocprofv3 summaries.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_final8; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -9 $OUT/pytest.log
WFA_TEST_OPTS=bt_stream_min=1,bt_stream_single=1 timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "parity_c3 or fuzz or other_penalties or chunks or mixed" > $OUT/pytest_stream.log 2>&1; echo "streamed backtrace forced: pytest rc $?" | tee -a $OUT/pytest_stream.log; tail -3 $OUT/pytest_stream.log
timeout 600 python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err
timeout 600 python bench.py --config c2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout 900 python bench.py --config c5s > $OUT/bench_c5s.json 2> $OUT/bench_c5s.err
for c in c3 c2 c5s; do python3 -c "
import json; d=json.load(open('$OUT/bench_$c.json')); c=d['config']; print('$c', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'fwd', round(c['main_kernel_ms'],3), 'h2h', c.get('host_to_host_ms'), 'packed', c.get('host_to_host_packed_ms'), 'align_us', c.get('single_pair_align_us'), 'frac', round(d['roofline']['frac'],4), 'stale', d['roofline'].get('traffic_stale'), 'cpu', d.get('cpu_baseline',{}).get('value'))"; done
timeout 900 bash scripts/profile_bench.sh r02_c3 > $OUT/prof_c3.log 2>&1
timeout 900 bash scripts/profile_bench.sh r02_c2 --config c2 > $OUT/prof_c2.log 2>&1
timeout 1200 bash scripts/profile_bench.sh r02_c5s --config c5s > $OUT/prof_c5s.log 2>&1
echo profiles done
