#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r02_b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_entries_gpu.py tests/test_cpp_host.py "tests/test_parity_gpu.py::test_blocked_kernel_arena_word_for_word" "tests/test_parity_gpu.py::test_failed_retry_pass_fails_the_call" "tests/test_parity_gpu.py::test_other_penalties" -m gpu -x -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -30 $OUT/pytest.log
hipcc -O2 --offload-arch=gfx950 -o /tmp/valu_rate2 scripts/probes/valu_rate2.hip 2> /dev/null
timeout 300 /tmp/valu_rate2 > $OUT/valu_rate2.txt 2>&1; echo "probe rc $?"
grep "waves/SIMD 4" $OUT/valu_rate2.txt
timeout 600 bash scripts/ab.sh --steps 10 --warmup 2 --host-entry 0 --latency 0 > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
# team kernel stamps on the c5s workload
mkdir -p /tmp/wfa_ts/wfa_amd/lib && cp wfa_amd/*.py /tmp/wfa_ts/wfa_amd/
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DWFA_TEAM_STAMPS -shared -o /tmp/wfa_ts/wfa_amd/lib/libwfahip.so wfa_amd/csrc/wfa_host.hip wfa_amd/csrc/wfa_gen.cpp wfa_amd/csrc/wfa_multi.cpp 2>/dev/null
cd /tmp/wfa_ts && timeout 300 python3 - > $REPO/$OUT/team_stamps.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, "/tmp/wfa_ts")
import wfa_amd as w
data = w.generate_pairs(5, 8, 100000, 0.10, n_threads=8)
al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
for rep in range(2):
    t0 = time.time(); r = al.align_arrays(*data); print("wall", time.time() - t0, al.last_timing(), flush=True)
PY
cd $REPO; tail -40 $OUT/team_stamps.txt
