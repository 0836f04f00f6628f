cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lane; rm -rf $OUT; mkdir -p $OUT
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --steps 1 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --opt ${LANE_OPT:-lane=2} > $OUT/pmc_$N.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_prof.py $OUT 2>&1 | grep "lane_kernel<false>\|blk_kernel" | grep "sum"
