#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_c; mkdir -p $OUT
WFAHIP_DEBUG_TIMING=1 timeout 300 python bench.py --steps 3 --warmup 2 --cpu-sample 0 --host-entry 0 --latency 0 --opt duo=1 > $OUT/bench_duo1.json 2> $OUT/bench_duo1.err
tail -30 $OUT/bench_duo1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --opt duo=1 > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
