#!/bin/bash
# A/B harness (GPU box) for the generic kernel: semi-global and wf-adaptive-off 1 kbp batches with each library
# variant under build/variants/.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
cp wfa_amd/lib/libwfahip.so /tmp/libwfahip.orig.so
for v in build/variants/*.so; do
  cp $v wfa_amd/lib/libwfahip.so; touch wfa_amd/lib/libwfahip.so
  for args in "--semi-global --pairs 100000" "--no-adaptive --pairs 100000 --opt pilot=0 --opt blk=0 --opt packed=0"; do
    python bench.py --cpu-sample 0 --steps 3 $args 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$(basename $v)', '$args', 'ms', round(d['ms_per_step'],3), 'ok', d['config']['status_ok'])"
  done
done
cp /tmp/libwfahip.orig.so wfa_amd/lib/libwfahip.so
