#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_entries_gpu.py "tests/test_parity_gpu.py::test_blocked_kernel_arena_word_for_word" -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
timeout 600 bash scripts/stamps.sh 400000 > $OUT/stamps.txt 2>&1; tail -12 $OUT/stamps.txt
