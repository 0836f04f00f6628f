#!/bin/bash
# round 5, first GPU call: the per-shape instances against the oracle, the whole -m gpu suite after the split into translation
# units, and first bench lines of c3 / p242
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_a; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_shapes_gpu.py -m gpu -q -x --durations=5 > $OUT/shapes.log 2>&1; echo "shapes rc $?" | tee -a $OUT/shapes.log; tail -15 $OUT/shapes.log
timeout 600 python bench.py --config p242 --steps 20 --warmup 3 --other-configs 0 --host-entry 0 --latency 0 > $OUT/bench_p242.json 2> $OUT/bench_p242.err; tail -c 1500 $OUT/bench_p242.json; tail -3 $OUT/bench_p242.err
timeout 600 python bench.py --steps 20 --warmup 3 --other-configs 0 --host-entry 0 --latency 0 --cpu-sample 0 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -c 600 $OUT/bench_c3.json
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_shapes_gpu.py --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log; tail -15 $OUT/pytest.log
