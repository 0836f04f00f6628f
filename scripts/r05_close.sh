#!/bin/bash
# closing run, second half: the whole suite at the closing sources, profiles, soak, configs[4] at its stated count
cd ${GRAFT_REPO_ROOT:-$(pwd)}
bash scripts/r05_final.sh r05_final tests lines profiles
timeout 900 python scripts/team_soak.py 120 team_xcd=2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_final/team_soak.txt | tail -3
timeout 1500 python scripts/c5_full.py 10000 500 > gpurun_out/r05_final/c5_full.json 2> gpurun_out/r05_final/c5_full.log; tail -2 gpurun_out/r05_final/c5_full.log; cat gpurun_out/r05_final/c5_full.json
timeout 1800 python scripts/soak.py > gpurun_out/r05_final/soak.log 2>&1; tail -3 gpurun_out/r05_final/soak.log
