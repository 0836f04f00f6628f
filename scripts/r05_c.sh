#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05_c; mkdir -p $OUT
WFA_TEST_OPTS=team_stripe=0 timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "team_kernel_wave_mode" > $OUT/wave_stripe0.log 2>&1; echo "wave_mode, stripe 0: rc $?"; tail -3 $OUT/wave_stripe0.log
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "team_kernel_stripe_mode or team_kernel_wavefronts or team_kernel_sees or team_kernel_paged" > $OUT/stripe.log 2>&1; echo "stripe tests rc $?"; tail -8 $OUT/stripe.log
timeout 900 bash scripts/team_stamps.sh 8 team_stripe=1 > $OUT/stamps1.txt 2>&1; grep -v "amdgpu.ids\|warning\|wfa_reg\|\^" $OUT/stamps1.txt | tail -30
timeout 900 bash scripts/team_stamps.sh 8 team_stripe=0 > $OUT/stamps0.txt 2>&1; grep -v "amdgpu.ids\|warning\|wfa_reg\|\^" $OUT/stamps0.txt | tail -30
