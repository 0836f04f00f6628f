#!/bin/bash
# The host entry (wfahip_align_batch: host blobs -> host results) of the headline workload with its slice timeline on stderr.  Usage (through gpurun): bash scripts/host_entry_timeline.sh
cd ${GRAFT_REPO_ROOT:-$(pwd)}
WFAHIP_DEBUG_TIMING=1 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --latency 0 --other-configs 0 2>&1 | grep -v "^{" | grep "wfahip\]" | tail -60
