#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_i; mkdir -p $OUT
timeout 600 bash scripts/ab.sh --steps 20 --warmup 2 --host-entry 0 --latency 0 > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "other_penalties or arena_word or synthetic or forward_kernel or short_read or known_answers or hand_over or full_size_parity" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
