#!/bin/bash
# Why wfa_wide_kernel hands pairs on (g3, one step): WFAHIP_WIDE_TRACE=1 makes the library print, per status and reason, how many pairs of the last chunk
# went to the ladder and the first few of them (score index, reason 1 = arena row does not fit / 2 = band outgrew the narrow phase's rings (+10: second launch), row width).
# Usage (through gpurun): bash scripts/wide_trace.sh
cd ${GRAFT_REPO_ROOT:-$(pwd)}
WFAHIP_WIDE_TRACE=1 timeout 300 python bench.py --cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 --config g3 --steps 1 --warmup 0 2>&1 | grep "wfahip\] wide" | sort | uniq -c | sort -rn | head -40
