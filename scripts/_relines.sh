cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_final3; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -c 600 $OUT/bench_c3.json
timeout 600 python bench.py --config c2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -c 900 $OUT/bench_c2.json
timeout 900 python bench.py --config c5s > $OUT/bench_c5s.json 2> $OUT/bench_c5s.err; tail -c 300 $OUT/bench_c5s.json
timeout 300 python bench.py --gpus 2 --share-gpus --steps 3 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 > $OUT/bench_share2.json 2> $OUT/bench_share2.err; tail -c 300 $OUT/bench_share2.json
python __graft_entry__.py smoke 2>&1 | tail -1
