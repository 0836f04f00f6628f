#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_f; mkdir -p $OUT
bash scripts/ab.sh --opt duo=1 --host-entry 0 --latency 0 --warmup 2 --steps 10 2>&1 | tee $OUT/ab.txt
WFA_OPTS=duo=1,blk_wide=0 timeout 600 bash scripts/stamps.sh 100000 > $OUT/stamps_duo.txt 2>&1; tail -8 $OUT/stamps_duo.txt
