cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lane2; rm -rf $OUT; mkdir -p $OUT
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_BRANCH" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --pairs 1024 --steps 1 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --cpu-all-cores 0 --opt lane=2 > $OUT/pmc_$N.log 2>&1
  tail -2 $OUT/pmc_$N.log | cut -c1-300
done
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_prof.py $OUT 2>&1 | grep "lane_kernel<false>" | grep "sum"
