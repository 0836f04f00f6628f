#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_g; mkdir -p $OUT
timeout 900 bash scripts/team_stamps.sh 8 > $OUT/stamps.txt 2>&1; grep "teamc [0-9]\] steps\|the workgroup in the middle\|exchange 1 in detail\|wall\|error" $OUT/stamps.txt | tail -25 | grep -A2 "stripe(team) [0-9][0-9][0-9][0-9][0-9]" | cut -c1-560
