cd $GRAFT_REPO_ROOT
TAG=r03_final4; OUT=gpurun_out/$TAG; mkdir -p $OUT
SOAK_SEEDS=107,108,109,116,117,118 timeout 900 python scripts/soak.py > $OUT/soak_short.log 2>&1; tail -2 $OUT/soak_short.log
timeout 900 bash scripts/profile_bench.sh ${TAG}_c3 > $OUT/prof_c3.log 2>&1
timeout 900 bash scripts/profile_bench.sh ${TAG}_c2 --config c2 > $OUT/prof_c2.log 2>&1
timeout 1500 bash scripts/profile_bench.sh ${TAG}_c5s --config c5s > $OUT/prof_c5s.log 2>&1
echo profiles done
