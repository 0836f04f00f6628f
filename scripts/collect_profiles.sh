#!/bin/bash
# Copy what a closing run (scripts/rNN_final.sh through gpurun) left under gpurun_out/ into profiles/ (tracked).
# Usage: scripts/collect_profiles.sh <run tag, e.g. r04_final2> <prefix, e.g. r04>
set -e
TAG=$1; PRE=$2
for d in gpurun_out/prof_${TAG}_*; do
  c=${d##*_${TAG}_}
  [ -f $d/summary.txt ] && cp $d/summary.txt profiles/${PRE}_${c}_summary.txt
  [ -f $d/pmc.json ] && cp $d/pmc.json profiles/${PRE}_${c}_pmc.json
done
mkdir -p profiles/${PRE}_bench_lines
for f in gpurun_out/$TAG/bench_*.json; do
  n=$(basename $f)
  # (one JSON line per file: anything a library printed before it is dropped)
  grep '^{' $f | tail -1 > profiles/${PRE}_bench_lines/${n%.json}_n1.json || true
done
[ -f gpurun_out/$TAG/align_latency.txt ] && cp gpurun_out/$TAG/align_latency.txt profiles/${PRE}_bench_lines/align_latency.txt
[ -f gpurun_out/$TAG/pytest.log ] && tail -15 gpurun_out/$TAG/pytest.log > profiles/${PRE}_gpu_tests.txt
[ -f gpurun_out/$TAG/pytest_poison.log ] && tail -3 gpurun_out/$TAG/pytest_poison.log >> profiles/${PRE}_gpu_tests.txt
[ -f gpurun_out/$TAG/soak.log ] && cp gpurun_out/$TAG/soak.log profiles/${PRE}_soak_log.txt
[ -f gpurun_out/$TAG/team_soak.txt ] && cat gpurun_out/$TAG/team_soak.txt >> profiles/${PRE}_team_xcd_soak.txt
ls profiles | grep "^${PRE}_" | head -40
