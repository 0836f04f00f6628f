#!/bin/bash
# GPU box: backtrace-kernel duration vs its occupancy cap (bt_lds option).  Usage: scripts/bt_sweep.sh <lds bytes>...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for l in "$@"; do
  OUT=$REPO/gpurun_out/bt_$l; mkdir -p $OUT
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 4 --warmup 1 --cpu-sample 0 --opt bt_lds=$l > $OUT/log 2>&1)
  f=$(ls $OUT/*/*kernel_stats.csv | head -1)
  echo "lds $l: $(grep -h 'backtrace\|blk_kernel<16' $f | awk -F, '{print $1, "calls", $2, "avg_us", $4/1000}' | tr '\n' ' ') $(grep -o '"ms_per_step": [0-9.]*' $OUT/log)"
done
