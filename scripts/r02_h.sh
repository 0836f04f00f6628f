#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_h; mkdir -p $OUT
timeout 600 bash scripts/ab.sh --steps 20 --warmup 2 --host-entry 0 --latency 0 > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
timeout 300 python bench.py --steps 20 --cpu-sample 0 > $OUT/bench_small.json 2>$OUT/bench_small.err; python3 -c "
import json; d=json.load(open('$OUT/bench_small.json')); c=d['config']; print('single_pair_align_us', c['single_pair_align_us'], 'h2h', c['host_to_host_ms'], 'packed', c.get('host_to_host_packed_ms'))"
timeout 300 python bench.py --config c2 --steps 200 --cpu-sample 0 > $OUT/bench_c2.json 2>$OUT/bench_c2.err; python3 -c "
import json; d=json.load(open('$OUT/bench_c2.json')); c=d['config']; print('c2 value', d['value'], 'single_pair_align_us', c['single_pair_align_us'], 'h2h', c['host_to_host_ms'])"
timeout 600 python -m pytest tests/test_entries_gpu.py tests/test_cli.py tests/test_cpp_host.py "tests/test_parity_gpu.py::test_known_answers" "tests/test_parity_gpu.py::test_ragged_and_edge_inputs" "tests/test_parity_gpu.py::test_reference_test_pairs_all_option_sets" -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
