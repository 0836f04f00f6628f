#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_p; mkdir -p $OUT
timeout 300 python scripts/host_entry_trace.py > $OUT/trace.txt 2>&1; cat $OUT/trace.txt | cut -c1-200 | tail -90
