#!/bin/bash
# Round 6, second half of the closing run (after scripts/r06_final.sh <tag> tests profiles stress + scripts/collect_profiles.sh + commit, so that the bench
# lines name PMC profiles of the same kernel sources): wfa_wide_kernel's tests once more, the bench lines + smoke, the parity soak on the shapes of the
# kernels this round touched (wfa_duo_kernel: global 1 kbp shapes; wfa_wide_kernel: semi-global), the team-kernel soak.  Usage (through gpurun): bash scripts/r06_close.sh
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout 240 -k "wide_kernel" 2>&1 | tail -3
bash scripts/r06_final.sh r06_final lines smoke
SOAK_SEEDS=101,104,110,113,123,124,125,126,127,128 timeout 1500 python scripts/soak.py > gpurun_out/r06_final/soak.log 2>&1; tail -4 gpurun_out/r06_final/soak.log | cut -c1-200
timeout 600 python scripts/team_soak.py 60 team_xcd=2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_final/team_soak.txt | tail -3
