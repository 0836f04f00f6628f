#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r02_j
timeout 300 python scripts/e0_debug.py > gpurun_out/r02_j/e0.txt 2>&1; cat gpurun_out/r02_j/e0.txt | cut -c1-400 | head -40
