#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_u; mkdir -p $OUT
cp wfa_amd/lib/libwfahip.so /tmp/libwfahip.orig.so
for v in build/variants/w4p4.so build/variants/w5p3.so; do
  cp $v wfa_amd/lib/libwfahip.so; touch wfa_amd/lib/libwfahip.so
  timeout 120 python bench.py --cpu-sample 0 --steps 10 --warmup 2 --host-entry 0 --latency 0 > $OUT/b.json 2> $OUT/b.err
  echo "rc $?"; python -c "
import json,sys
d=json.load(open('$OUT/b.json'))
print('$(basename $v)', 'fwd_ms', round(d['config']['main_kernel_ms'],3), 'all_ms', round(d['config']['kernel_ms_per_step'],3), 'pairs/s', round(d['value']), 'ok', d['config']['status_ok'], 'retry', d['config']['retried_pairs'])" || tail -3 $OUT/b.err
done
cp /tmp/libwfahip.orig.so wfa_amd/lib/libwfahip.so
