cd $GRAFT_REPO_ROOT
TAG=r03_final3; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "lane" > $OUT/pytest_lane.log 2>&1; tail -2 $OUT/pytest_lane.log
timeout 900 bash scripts/profile_bench.sh ${TAG}_c3 > $OUT/prof_c3.log 2>&1
timeout 900 bash scripts/profile_bench.sh ${TAG}_c2 --config c2 > $OUT/prof_c2.log 2>&1
timeout 1500 bash scripts/profile_bench.sh ${TAG}_c5s --config c5s > $OUT/prof_c5s.log 2>&1
echo profiles done
