#!/bin/bash
# A/B harness (GPU box): run the bench with each library variant under build/variants/ (made by scripts/mkvariant.sh).
# Usage: scripts/ab.sh [bench args...]   -- prints forward-kernel ms per variant
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
cp wfa_amd/lib/libwfahip.so /tmp/libwfahip.orig.so
for v in build/variants/*.so; do
  cp $v wfa_amd/lib/libwfahip.so; touch wfa_amd/lib/libwfahip.so
  python bench.py --cpu-sample 0 --steps 3 "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$(basename $v)', 'fwd_ms', round(d['config']['main_kernel_ms'],3), 'all_ms', round(d['config']['kernel_ms_per_step'],3), 'pairs/s', round(d['value']), 'ok', d['config']['status_ok'], 'retry', d['config']['retried_pairs'])"
done
cp /tmp/libwfahip.orig.so wfa_amd/lib/libwfahip.so
