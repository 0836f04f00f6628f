#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_d; mkdir -p $OUT
WFA_OPTS=duo=1,blk_wide=0 timeout 600 bash scripts/stamps.sh 400000 > $OUT/stamps_duo.txt 2>&1; tail -12 $OUT/stamps_duo.txt
timeout 900 bash scripts/profile_bench.sh r03_duo1 --opt duo=1 > $OUT/prof.log 2>&1; tail -40 $OUT/prof.log
