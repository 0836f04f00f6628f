#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_g; mkdir -p $OUT
for t in 512 256; do
WFA_EXTRA="-DWFA_TC_THREADS=$t" timeout 900 bash scripts/team_stamps.sh 8 > $OUT/stamps_t$t.txt 2>&1; echo "== threads $t"; grep "teamc 0\|teamc 7\|wall\|error" $OUT/stamps_t$t.txt | cut -c1-700 | tail -6
done
