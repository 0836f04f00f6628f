#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for c in g_T2_team s_T2_wave; do
  timeout 120 python scripts/teamc_probe.py $c 2>&1 | grep -v "amdgpu.ids\|loop not unrolled\|wfa_reg_kernel\|\^\|warning generated\|In file included\|XCC ids" | tail -12
done
