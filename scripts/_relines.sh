cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_final4; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -c 300 $OUT/bench_c3.json
timeout 600 python bench.py --config c2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -c 300 $OUT/bench_c2.json
timeout 900 python bench.py --config c5s > $OUT/bench_c5s.json 2> $OUT/bench_c5s.err; tail -c 200 $OUT/bench_c5s.json
