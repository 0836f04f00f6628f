#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_f; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q --durations=6 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
timeout 900 bash scripts/profile_bench.sh r02_c3 > $OUT/prof_c3.log 2>&1; tail -3 $OUT/prof_c3.log
timeout 900 bash scripts/profile_bench.sh r02_c2 --config c2 > $OUT/prof_c2.log 2>&1; tail -3 $OUT/prof_c2.log
timeout 1200 bash scripts/profile_bench.sh r02_c5s --config c5s > $OUT/prof_c5s.log 2>&1; tail -3 $OUT/prof_c5s.log
