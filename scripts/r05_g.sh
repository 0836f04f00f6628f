#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_g; mkdir -p $OUT
for t in 512; do
WFA_EXTRA="-DWFA_TC_THREADS=$t" timeout 900 bash scripts/team_stamps.sh 8 > $OUT/stamps_t$t.txt 2>&1; echo "== threads $t"; grep "teamc [0-9]\] steps\|the workgroup in the middle\|exchange 1 in detail\|^wall" $OUT/stamps_t$t.txt | tail -25 | grep -A2 "stripe(team) 32274\|^wall" | cut -c1-560
done
