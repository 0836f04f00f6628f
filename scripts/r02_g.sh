#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_g; mkdir -p $OUT
timeout 1500 python scripts/soak.py > $OUT/soak.txt 2>&1; echo "soak rc $?" >> $OUT/soak.txt; tail -22 $OUT/soak.txt
timeout 900 python scripts/soak_chunks.py > $OUT/soak_chunks.txt 2>&1; echo "soak_chunks rc $?" >> $OUT/soak_chunks.txt; tail -5 $OUT/soak_chunks.txt
timeout 900 python scripts/fuzz_many.py 5000 150 > $OUT/fuzz.txt 2>&1; echo "fuzz rc $?" >> $OUT/fuzz.txt; tail -3 $OUT/fuzz.txt
timeout 600 python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -c 2500 $OUT/bench_c3.json
