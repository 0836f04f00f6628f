#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_c; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout 300 python bench.py --steps 40 --warmup 3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -c 1500 $OUT/bench_c3.json
timeout 600 python bench.py --config c5s --steps 3 --warmup 2 > $OUT/bench_c5s.json 2> $OUT/bench_c5s.err; tail -c 1800 $OUT/bench_c5s.json
timeout 900 bash scripts/profile_bench.sh r02_c3 > $OUT/prof_c3.log 2>&1; tail -5 $OUT/prof_c3.log
