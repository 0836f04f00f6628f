cd $GRAFT_REPO_ROOT
python3 scripts/_one_pair.py
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_one; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/scripts/_one_pair.py > $OUT/log.txt 2>&1
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
