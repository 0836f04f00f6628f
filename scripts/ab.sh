#!/bin/bash
# A/B harness (GPU box): run the bench with each library variant under build/variants/ (made by scripts/mkvariant.sh),
# the shipped library first.  Usage: scripts/ab.sh [bench args...]   -- prints forward-kernel / step ms per variant
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
cp wfa_amd/lib/libwfahip.so /tmp/libwfahip.orig.so
mkdir -p gpurun_out
for v in /tmp/libwfahip.orig.so build/variants/*.so; do
  [ -f $v ] || continue
  case $v in *stamps*) continue;; esac
  cp $v wfa_amd/lib/libwfahip.so; touch wfa_amd/lib/libwfahip.so
  python bench.py --cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 --steps 10 --warmup 2 "$@" 2> gpurun_out/ab_$(basename $v).err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$(basename $v)', 'fwd_ms', round(d['config']['main_kernel_ms'],3), 'all_ms', round(d['config']['kernel_ms_per_step'],3), 'step_ms', round(d['ms_per_step'],3), 'pairs/s', round(d['value']), 'ok', d['config']['status_ok'], 'retry', d['config']['retried_pairs'])" || tail -3 gpurun_out/ab_$(basename $v).err
done
cp /tmp/libwfahip.orig.so wfa_amd/lib/libwfahip.so
