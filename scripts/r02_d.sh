#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_d; mkdir -p $OUT
timeout 600 python -m pytest "tests/test_parity_gpu.py::test_blocked_kernel_arena_word_for_word" -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 bash scripts/ab.sh --steps 20 --warmup 2 --host-entry 0 --latency 0 > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
# parity of the asm variant on the arena test too
cp wfa_amd/lib/libwfahip.so /tmp/orig.so; cp build/variants/v2asm.so wfa_amd/lib/libwfahip.so
timeout 600 python -m pytest "tests/test_parity_gpu.py::test_blocked_kernel_arena_word_for_word" "tests/test_parity_gpu.py::test_synthetic_batches" "tests/test_parity_gpu.py::test_forward_kernel_variants" -m gpu -x -q > $OUT/pytest_asm.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_asm.log
tail -5 $OUT/pytest_asm.log
cp /tmp/orig.so wfa_amd/lib/libwfahip.so
