cd ${GRAFT_REPO_ROOT:-$(pwd)}
WFAHIP_WIDE_TRACE=1 timeout 300 python bench.py --cpu-sample 0 --host-entry 0 --latency 0 --other-configs 0 --config g3 --steps 1 --warmup 0 2>&1 | grep "wfahip\] wide" | sort | uniq -c | sort -rn | head -40
