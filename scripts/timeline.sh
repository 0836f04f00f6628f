#!/bin/bash
# GPU box: kernel timeline of the last bench step.  Usage: scripts/timeline.sh <tag> [bench args...]
TAG=${1:-tl}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 "$@" > $OUT/log 2>&1
python3 $REPO/scripts/timeline.py $OUT
tail -1 $OUT/log | cut -c1-300
