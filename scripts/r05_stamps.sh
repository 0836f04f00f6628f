#!/bin/bash
# the in-kernel phase table of wfa_teamc_kernel (diagnostic build, scripts/team_stamps.sh) on the configs[4] sample and on 32 pairs
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_stamps; mkdir -p $OUT
timeout 900 bash scripts/team_stamps.sh 8 > $OUT/stamps8.txt 2>&1
timeout 900 bash scripts/team_stamps.sh 8 team_pipe=0 > $OUT/stamps8_nopipe.txt 2>&1
timeout 1200 bash scripts/team_stamps.sh 32 > $OUT/stamps32.txt 2>&1
for f in stamps8 stamps8_nopipe stamps32; do echo "== $f"; grep "teamc [0-9]*\] \|^wall" $OUT/$f.txt | tail -40 | cut -c1-200 | tail -6; done
