#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_o; mkdir -p $OUT
bash scripts/ab.sh --host-entry 0 --latency 0 --warmup 2 --steps 10 2>&1 | tee $OUT/ab.txt
