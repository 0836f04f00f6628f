#!/bin/bash
# A/B of library options (GPU box): one short bench run per option string, each under its own timeout.
# Usage: scripts/ab_opts.sh "<bench args>" "opt string 1" "opt string 2" ...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out
BA=$1; shift
for o in "$@"; do
  echo -n "== [$o] "
  timeout ${ABTIMEOUT:-150} python bench.py --cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 --steps 10 --warmup 2 $BA $o 2> gpurun_out/ab_opts.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('fwd_ms', round(c['main_kernel_ms'],3), 'all_ms', round(c['kernel_ms_per_step'],3), 'step_ms', round(d['ms_per_step'],3), 'ok', c['status_ok'], 'retry', c['retried_pairs'])" || { echo FAILED; tail -3 gpurun_out/ab_opts.err; }
done
