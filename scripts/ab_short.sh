#!/bin/bash
# A/B harness (GPU box) for short reads (BASELINE configs[1]): 1e5 and 1e6 x 150 bp @2 %, wf-adaptive off, with each
# library variant under build/variants/.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
cp wfa_amd/lib/libwfahip.so /tmp/libwfahip.orig.so
for rep in 1 2; do
for v in build/variants/*.so; do
  cp $v wfa_amd/lib/libwfahip.so; touch wfa_amd/lib/libwfahip.so
  for n in 100000 1000000; do
    python bench.py --cpu-sample 0 --steps 30 --pairs $n --length 150 --error 0.02 --seed 2 --no-adaptive "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$(basename $v)', 'n=$n', 'ms', round(d['ms_per_step'],4), 'fwd', round(d['config']['main_kernel_ms'],4), 'ok', d['config']['status_ok'])"
  done
done
done
cp /tmp/libwfahip.orig.so wfa_amd/lib/libwfahip.so
