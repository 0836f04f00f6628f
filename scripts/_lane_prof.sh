cd $GRAFT_REPO_ROOT
for o in lane=0 lane=2; do python3 bench.py --config c2 --cpu-sample 0 --host-entry 0 --latency 0 --opt $o 2>&1 | tail -1 > gpurun_out/lane_c2_$o.json; done
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lane; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --steps 3 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --opt lane=2 > $OUT/stats.log 2>&1
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --steps 1 --warmup 1 --cpu-sample 0 --host-entry 0 --latency 0 --opt lane=2 > $OUT/pmc_$N.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat gpurun_out/lane_c2_*.json
head -40 $OUT/summary.txt
