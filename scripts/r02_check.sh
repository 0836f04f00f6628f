#!/bin/bash
# GPU box: the -m gpu suite, then the three bench configurations, the VALU-rate probe; logs under gpurun_out/r02_check
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_check; mkdir -p $OUT
nproc > $OUT/nproc.txt; free -g >> $OUT/nproc.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -c 3000 $OUT/bench_c3.json
timeout 600 python bench.py --config c2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -c 2500 $OUT/bench_c2.json
timeout 900 python bench.py --config c5s > $OUT/bench_c5s.json 2> $OUT/bench_c5s.err; tail -c 2500 $OUT/bench_c5s.json
hipcc -O2 --offload-arch=gfx950 -o /tmp/valu_rate scripts/probes/valu_rate.hip 2> $OUT/probe_build.err
for op in 0 23 1 36 24 28 34 38 2 22 3 19 21 30 31; do for w in 2 4 8; do /tmp/valu_rate $op $w; done; done > $OUT/valu_rate.txt 2>&1
cat $OUT/valu_rate.txt
